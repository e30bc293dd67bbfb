#!/bin/bash
# seg_reduce tuning variants (RELGNN_SEG_VARIANT bits: 1 = unroll16, 2 = non-temporal streams, 4 = no XCD swizzle; 8 = unroll 4)
for v in 0 1 2 3 4 5 6 7 8; do
  echo -n "variant $v: "; RELGNN_SEG_VARIANT=$v python scripts/kernel_only.py 100 2>/dev/null
done
