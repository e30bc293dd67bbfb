"""tf_gnn_samples_amd.config: every route switch in one place, read at call time; the environment only supplies initial values."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_override_sets_validates_and_restores():
    from tf_gnn_samples_amd import config
    before = config.current()
    with config.override(gemm="lib", limb="triple") as s:
        assert s.gemm == "lib" and not s.limb_gemm and not s.limb_pair and not s.bwd_overlap_on
        with config.override(gemm="limb"):
            assert config.settings.limb_gemm and not config.settings.limb_pair and config.settings.bwd_overlap_on
        assert config.settings.gemm == "lib"
    assert config.current() == before
    with pytest.raises(ValueError, match="RELGNN_GEMM must be one of"):
        with config.override(gemm="fast"):
            pass
    with pytest.raises(AttributeError, match="no switch"):
        with config.override(gem="lib"):
            pass
    with pytest.raises(ValueError):
        config.settings.limb = "quad"
    try:
        with config.override(tn="lib"):
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert config.current() == before


def test_the_environment_supplies_the_initial_values_only():
    code = ("from tf_gnn_samples_amd import config as c; import os; os.environ['RELGNN_GEMM'] = 'panel'; "
            "print(c.settings.gemm, c.settings.limb, c.settings.edge_bwd, c.settings.pair_tables)")
    env = dict(os.environ, RELGNN_GEMM="lib", RELGNN_LIMB="triple", RELGNN_EDGE_BWD_REGATHER="1", PYTHONPATH=str(ROOT))
    env.pop("RELGNN_EDGE_BWD", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert out == ["lib", "triple", "regather", "auto"]
    bad = subprocess.run([sys.executable, "-c", "import tf_gnn_samples_amd.config"], env=dict(env, RELGNN_TN="fast"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert bad.returncode != 0 and "RELGNN_TN must be one of" in bad.stderr


def test_environment_values_of_earlier_rounds_are_still_accepted():
    """RELGNN_PAIR_TABLES=true / false / '' (and the other 0 / 1 switches) were accepted before the switches moved into
    config; a removed variable (RELGNN_PAIR_CHUNK) is reported once instead of silently ignored."""
    code = ("import warnings; warnings.simplefilter('always'); from tf_gnn_samples_amd import config as c; "
            "print(c.settings.pair_tables, c.settings.limb_cut, c.settings.weight_limb_cache, c.settings.bwd_overlap, c.settings.gemm)")
    env = dict(os.environ, RELGNN_PAIR_TABLES="true", RELGNN_LIMB_CUT="False", RELGNN_WEIGHT_LIMB_CACHE="", RELGNN_BWD_OVERLAP="off",
               RELGNN_GEMM="LIB", RELGNN_PAIR_CHUNK="256", PYTHONPATH=str(ROOT))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    assert r.stdout.split() == ["1", "0", "1", "0", "lib"]
    assert "RELGNN_PAIR_CHUNK is set and has no effect" in r.stderr
    bad = subprocess.run([sys.executable, "-c", "import tf_gnn_samples_amd.config"], env=dict(env, RELGNN_EDGE_BWD="maybe"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert bad.returncode != 0 and "RELGNN_EDGE_BWD must be one of" in bad.stderr


def test_the_default_limb_arithmetic_is_the_exact_split():
    """Round 5: the headline runs on the default switches, and the default of the tall Dense products is the exact three-bf16-limb
    split (fp32 semantics); the 22-bit two-fp16-limb form is opt-in."""
    from tf_gnn_samples_amd import config
    assert config.default_of("gemm") == "limb" and config.default_of("limb") == "triple"
    with config.override(limb="pair"):
        assert config.settings.limb_pair
    assert not config.settings.limb_pair or config.settings.limb == "pair"


def test_nothing_else_in_the_package_reads_relgnn_environment_variables():
    """One table, one reader: a RELGNN_* name anywhere else in the Python package is a comment / docstring, never os.environ."""
    offenders = []
    for path in (ROOT / "tf_gnn_samples_amd").rglob("*.py"):
        if path.name == "config.py":
            continue
        for i, line in enumerate(path.read_text().splitlines(), 1):
            if "environ" in line and "RELGNN_" in line:
                offenders.append("%s:%d" % (path.relative_to(ROOT), i))
    assert not offenders, offenders


def test_the_documented_switch_table_is_the_code_s_table():
    """README.md prints config.describe(); a switch added to the code must be added there (and vice versa)."""
    from tf_gnn_samples_amd import config
    text = (ROOT / "README.md").read_text()
    documented = set(re.findall(r"`(RELGNN_[A-Z_0-9]+)`", text[text.index("## Switches"):]))
    in_code = {row[0] for row in config.describe()}
    assert in_code <= documented, sorted(in_code - documented)
    for env, name, default, allowed, doc in config.describe():
        assert config.attribute_of(env) == name and config.default_of(name) == default and doc


def test_driver_entry_points_reference_only_names_that_exist():
    """smoke() and the bench scripts run on the GPU box only: a module attribute renamed under them (round 4: dense._LIMB_GEMM ->
    config.settings.limb_gemm) must fail HERE, not at the end of a round."""
    import importlib
    aliases = {"DN": "tf_gnn_samples_amd.dense", "ops": "tf_gnn_samples_amd.ops", "config": "tf_gnn_samples_amd.config",
               "route_config": "tf_gnn_samples_amd.config", "_lib": "tf_gnn_samples_amd._lib"}
    missing = []
    for name in ("__graft_entry__.py", "bench.py", "bench_other.py", "bench_roofline.py", "scripts/exp_trajectory_routes.py",
                 "scripts/bench_limb_gemm.py", "scripts/exp_gemm_after_gather.py"):
        text = (ROOT / name).read_text()
        for alias, module in aliases.items():
            if alias == "config" and name.startswith("bench"):       # (bench.py imports it as route_config; "config.x" there is JSON prose)
                continue
            mod = importlib.import_module(module)
            for attr in set(re.findall(r"(?<![\w.])%s\.([A-Za-z_]\w*)" % alias, text)):
                if not hasattr(mod, attr):
                    missing.append("%s: %s.%s" % (name, alias, attr))
    assert not missing, missing
