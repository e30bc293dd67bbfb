cd $GRAFT_REPO_ROOT
for v in 0 128 64; do
  RELGNN_PANEL_NC=$v timeout 300 python bench_other.py C5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('C5 panel_nc=$v fixed batch', d['train_ms'], d['fwd_ms'])"
  RELGNN_PANEL_NC=$v RELGNN_BWD_OVERLAP=0 timeout 300 python bench_other.py C5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('C5 panel_nc=$v no overlap fixed batch', d['train_ms'], d['fwd_ms'])"
done
