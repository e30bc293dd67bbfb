"""Every route switch of the package in ONE place.

The reference has no such switches: each one here selects between two implementations of the SAME function (same results up to
the rounding stated in DESIGN.md), kept because both were measured.  `settings` is read at call time, so a value can be changed
for a region of code in one process:

    from tf_gnn_samples_amd import config
    with config.override(gemm="lib"):
        ...

The RELGNN_* environment variables only supply the initial values (read once, when this module is imported); nothing else in the
package reads the environment.  `describe()` returns the table that README.md / INTEGRATION.md print and that
tests/test_config_cpu.py checks against this file.
"""
import contextlib
import os
from typing import Dict, Iterator, List, Tuple

# attribute -> (environment variable, default, allowed values or None, what it selects)
_SPEC: Dict[str, Tuple[str, str, Tuple[str, ...], str]] = {
    "gemm": ("RELGNN_GEMM", "limb", ("limb", "lib", "panel", "torch"),
             "route of the node-side Dense products: fp32 from bf16 / fp16 limbs on the 16-bit matrix pipe (csrc/limb_gemm.hip) | "
             "exact fp32 through hipBLASLt with cached solutions | the exact-fp32 row-panel MFMA kernel | torch.mm"),
    "limb": ("RELGNN_LIMB", "triple", ("triple", "pair"),
             "limb arithmetic of the aggregate-first layer's three products: three bf16 limbs everywhere (the EXACT split "
             "hi + mid + lo == x: fp32 semantics) | two fp16 limbs behind power-of-two scales (22-bit operands: a reduced-precision "
             "fast path, opt-in)"),
    "limb_pair_parts": ("RELGNN_LIMB_PAIR_PARTS", "nn,nt,tn", None,
                        "which of the aggregate-first layer's products take the two-limb form (diagnostic): forward nn, input "
                        "gradient nt, weight gradient tn"),
    "limb_pc": ("RELGNN_LIMB_PC", "fwd", ("fwd", "0", "1"),
                "exact-split products with N = 256 or K <= 256: producer / matrix wave roles with LDS hand-over "
                "(csrc/limb_gemm_pc.hip) for the forward products only (input-gradient products run next to the side stream's weight "
                "gradient, which a kernel that holds every CU starves) | never | for every product that fits; the same bits"),
    "typed_pc": ("RELGNN_TYPED_PC", "0", ("0", "fwd", "1"),
                 "per-(node, type) products of many-type graphs (K, N in {128, 256}): the LDS-resident-weights panel kernels | the "
                 "wave-role kernel (csrc/limb_gemm_pc_typed.hip: gathering producer waves + barrier-free matrix waves, every row "
                 "gathered once) for the gathered N = 256 forward product | for every typed product it takes.  The same bits; alone "
                 "330-340 vs 402-407 us (forward N = 256), 202 vs 222 (forward N = 128), 234 vs 230 and 350 vs 366 (input "
                 "gradients); on the C5 step no difference outside the noise (33.0-34.0 ms on all three values), hence off"),
    "gru_cell": ("RELGNN_GRU_CELL", "1", ("0", "1"),
                 "the GRU cell forward of a GGNN layer (128 units over 128-wide messages): three limb products + the gate kernel + "
                 "the output kernel | ONE wave-role kernel per pass (csrc/gru_cell.hip): both products, gates, candidate and blend, r * h "
                 "handed to the candidate's k-loop through LDS, and the data path of the backward; x @ kernel + h @ recurrent_kernel is one "
                 "accumulation there: the last bits differ from the composition"),
    "limb_cut": ("RELGNN_LIMB_CUT", "1", ("0", "1"),
                 "N % 128 >= 96 products (the 121 logits of the PPI head) on the 128-column limb panels with the last chunk cut at N"),
    "head_pad": ("RELGNN_HEAD_PAD", "1", ("0", "1"),
                 "PPI head backward: the loss gradient written into rows zero-padded to a multiple of 16 columns, so that the head's "
                 "input-gradient product (K = 121) runs on the limb route with the last layer's ReLU' in its epilogue | the library "
                 "product + a ReLU' pass"),
    "weight_limb_cache": ("RELGNN_WEIGHT_LIMB_CACHE", "1", ("0", "1"),
                          "limb images of the weights kept across the products of a step (re-split once after the optimizer's update)"),
    "act_fusion": ("RELGNN_ACT_FUSION", "1", ("0", "1"),
                   "activations of the driver loop's Dense layers in the product's epilogue and activation GRADIENTS folded into the "
                   "input-gradient product of the layer above (relgnn_limb_gemm_xf32_dact) | separate passes"),
    "tn": ("RELGNN_TN", "stream", ("stream", "lib"),
           "weight gradients with outputs up to 256 x 256: the streaming MFMA kernel (csrc/gemm_tn_stream.hip) | library split-K"),
    "rgcn_order": ("RELGNN_RGCN_ORDER", "aggregate_first", ("aggregate_first", "transform_first"),
                   "sum / mean / sqrt_n RGCN layers: gather raw states into the (target, type) buckets, then one K = L*D product | "
                   "the reference's order (per-type transform, then gather)"),
    "rgcn_fused": ("RELGNN_RGCN_FUSED", "0", ("0", "1"),
                   "aggregate-first RGCN layer, 256 -> 256 states, exact-split arithmetic: gather and product in ONE kernel "
                   "(csrc/rgcn_fused.hip: gather waves feed the MFMA waves through LDS, bit-identical) | gather kernel, then product"),
    "agg_acc": ("RELGNN_AGG_ACC", "f32", ("f32", "f64"),
                "accumulator width of the bucket sums in front of the aggregate-first product"),
    "bwd_overlap": ("RELGNN_BWD_OVERLAP", "auto", ("auto", "0", "1"),
                    "aggregate-first backward: the weight gradient on a side stream next to the input gradient's gather "
                    "(auto: on with gemm=limb, off otherwise)"),
    "edge_bwd": ("RELGNN_EDGE_BWD", "auto", ("auto", "emit", "regather"),
                 "FiLM / pair kernels, gradient of the gathered rows: per-message gradients written and gather-reduced | the "
                 "by-source pass re-gathers (auto: emit on compact pair tables or D <= 128)"),
    "edge_sign_mask": ("RELGNN_EDGE_SIGN_MASK", "0", ("0", "1"),
                       "FiLM regather backward: pass A leaves one sign bit per message and feature for pass B"),
    "typed": ("RELGNN_TYPED", "panel", ("panel", "bmm"),
              "per-(node, type) transforms of many-type graphs: one gathered-row MFMA launch | index_select + torch.bmm"),
    "typed_tn": ("RELGNN_TYPED_TN", "auto", ("auto", "limb", "panel"),
                 "typed weight-gradient partials of many-type graphs: the gathered three-limb TN kernel on the 16-bit matrix pipe "
                 "(needs gemm=limb) | the exact-fp32 row-panel MFMA kernel (auto: limb for 256-column outputs, where it measured "
                 "405 vs 484 us; panel for 128-column ones, 262 vs 325 us: profiles/r05_typed_tn.jsonl)"),
    "pair_tables": ("RELGNN_PAIR_TABLES", "auto", ("auto", "0", "1"),
                    "compact tables over the non-empty (node, type) buckets (auto: L >= 8 and < 60 % of the buckets non-empty)"),
    "rgat_fused_sums": ("RELGNN_RGAT_FUSED_SUMS", "1", ("0", "1"),
                        "RGAT backward: the two score-table gradients carried along by the dz pass and the by-source gather"),
    "allreduce": ("RELGNN_ALLREDUCE", "flat", ("flat", "overlap"),
                  "data-parallel gradient all-reduce: one flat collective after the backward | buckets launched during the backward"),
}


class _Settings:
    __slots__ = tuple(_SPEC)

    def __init__(self):
        for name, (env, default, allowed, _) in _SPEC.items():
            value = os.environ.get(env, default)
            if name == "edge_bwd" and env not in os.environ and os.environ.get("RELGNN_EDGE_BWD_REGATHER") is not None:
                value = "regather"                       # (the older spelling of RELGNN_EDGE_BWD=regather)
            if allowed is not None and value not in allowed:
                value = _legacy_spelling(value, allowed, default)
            _check(name, value)
            object.__setattr__(self, name, value)
        for env, why in _REMOVED.items():
            if env in os.environ:
                import warnings
                warnings.warn("%s is set and has no effect any more: %s" % (env, why), stacklevel=3)

    def __setattr__(self, name, value):
        _check(name, value)
        object.__setattr__(self, name, value)

    # ---- derived ----
    @property
    def limb_gemm(self) -> bool:
        return self.gemm == "limb"

    @property
    def limb_pair(self) -> bool:
        return self.gemm == "limb" and self.limb == "pair"

    @property
    def bwd_overlap_on(self) -> bool:
        return self.bwd_overlap == "1" or (self.bwd_overlap == "auto" and self.gemm == "limb")

    def pair_part(self, kind: str) -> bool:
        return kind in self.limb_pair_parts.split(",")


# switches that existed in earlier rounds and are gone (the environment variable is ignored: say so once, at import)
_REMOVED = {
    "RELGNN_PAIR_CHUNK": "compact pair tables always use 512-row tiles (graph.PAIR_CHUNK)",
    "RELGNN_FEATURE_PAD": "zero-padded feature rows for the input projection measured slower (1.996 vs 1.937 ms per C2 step, round 5) "
                          "and were removed in round 6",
    "RELGNN_ASSEMBLE_STREAM": "resident folds assemble the next batch on the caller's stream (the side-stream form stretched the GEMM "
                              "it met from 113 to 229 us: 2.44 vs 2.38 ms, round 2; removed in round 6)",
}


def _legacy_spelling(value: str, allowed: Tuple[str, ...], default: str) -> str:
    """Environment values that the switches accepted before they moved here: true / false / yes / no / on / off (any case) for
    the 0 / 1 switches, and the empty string for 'leave the default'.  Anything else is returned unchanged (and rejected)."""
    v = value.strip().lower()
    if v == "":
        return default
    if "0" in allowed and "1" in allowed:
        if v in ("true", "yes", "on"):
            return "1"
        if v in ("false", "no", "off"):
            return "0"
    return v if v in allowed else value


def _check(name: str, value: str) -> None:
    if name not in _SPEC:
        raise AttributeError("tf_gnn_samples_amd.config: no switch %r (known: %s)" % (name, ", ".join(_SPEC)))
    env, _, allowed, _ = _SPEC[name]
    if not isinstance(value, str):
        raise ValueError("%s: switch values are strings (got %r)" % (env, value))
    if allowed is not None and value not in allowed:
        raise ValueError("%s must be one of %s (got %r)" % (env, ", ".join(allowed), value))


settings = _Settings()


@contextlib.contextmanager
def override(**values: str) -> Iterator[_Settings]:
    """Set switches for the duration of a `with` block (validated; restored on exit, also on exceptions).  Limb images of weights
    cached under one arithmetic are keyed by it, so switching needs no cache flush."""
    old = {}
    try:
        for name, value in values.items():
            _check(name, value)
            old[name] = getattr(settings, name)
            setattr(settings, name, value)
        yield settings
    finally:
        for name, value in old.items():
            setattr(settings, name, value)


def describe() -> List[Tuple[str, str, str, str, str]]:
    """(environment variable, attribute, default, allowed values, meaning) per switch."""
    return [(env, name, default, " | ".join(allowed) if allowed else "comma-separated subset of nn,nt,tn", doc)
            for name, (env, default, allowed, doc) in _SPEC.items()]


def attribute_of(env: str) -> str:
    """'RELGNN_GEMM' -> 'gemm'."""
    for name, spec in _SPEC.items():
        if spec[0] == env:
            return name
    raise KeyError("no switch reads %s" % env)


def default_of(name: str) -> str:
    return _SPEC[name][1]


def current() -> Dict[str, str]:
    return {name: getattr(settings, name) for name in _SPEC}


if __name__ == "__main__":                     # the markdown table of README.md "Switches"
    print("| Environment variable | `config.settings.` | default | values | selects |")
    print("|---|---|---|---|---|")
    for env, name, default, allowed, doc in describe():
        print("| `%s` | `%s` | `%s` | %s | %s |" % (env, name, default, allowed.replace(" | ", ", "), doc.replace(" | ", " / ")))
