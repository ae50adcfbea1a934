"""ONE switch for everything that is not the product path.

The product has six environment variables (README.md): ``UR_LIB_PATH``, ``UR_IGEMM_TUNING``, ``UR_PRECISE_RESIDUAL``,
``UR_HOIST``, ``UR_WGRAD_PENDING_MB`` and the smoke test's ``UR_SMOKE_CONFIG``; two more are read by opt-in EXPERIMENT BUILDS of
the native library only (``UR_DXS=1`` with ``make DXS=1``, csrc/igemm.hip; ``UR_WGRAD_ABLATE`` with ``make WGRAD_ABL=1``,
csrc/wgrad.hip) and do nothing in the product build.  Every other toggle -- the A/B off-switches
of accepted optimisations and the on-switches of paths that were built, measured slower and left in for the record
("tried and rejected") -- is an entry of

    UR_EXPERIMENT="name,no_name,name=value,..."        e.g.  UR_EXPERIMENT="no_tchain,side_stream=2,splitk_gn"

``flag(name, default)`` is True for ``name``, False for ``no_name``; ``number(name, default)`` reads ``name=value``.  Read
once at import; unknown names raise at the first query of any flag so that a typo cannot silently run the default."""
from __future__ import annotations

import os

KNOWN = {
    # ---- off-switches of accepted paths (A/B runs) ----
    "tchain": "row-local chain kernels at the 320-channel level (fused.py)",
    "qkv_one_launch": "q | k | v as one GEMM with a transposed value output (fused.py)",
    "ctxkv_one_launch": "prompt K | V^T of a phase as one GEMM (fused.py)",
    "vt_first": "V^T projection issued before the q | k projection (only without qkv_one_launch)",
    "exchange_early": "exchange GEMMs right behind the kernel that produced their skip (fused.py)",
    "ctx3_early": "up-phase time / prompt projections issued before phase 1 (fused.py)",
    "fold_shortcut": "resnet conv_shortcut as the 1x1 K tail of conv2 (ops.py)",
    "multi_transpose": "batched weight transposes of the backward (backward.py)",
    "batch_wt": "all W^T of a network in one launch group (backward.py)",
    "fused_gradnorm": "clipping norm collected where the gradients are written (backward.py)",
    "wgrad": "weight gradients on ur_wgrad instead of transposed-operand GEMMs (backward.py)",
    "wgrad_table": "measured (tile, slices) table of ur_wgrad (backward.py)",
    "wgrad_defer": "weight gradients deferred to the end of a network's backward and grouped (backward.py)",
    "norm_defer": "gamma / beta gradient sums deferred to the same barriers (backward.py)",
    "heads_multi": "split / merge heads of several tensors in one launch (backward.py)",
    "flash_backward": "flash attention backward instead of the materialised-P path (backward.py)",
    "forward_lse": "the forward attention hands its row log-sum-exp to the backward (backward.py)",
    "batch_casts": "fp32 -> bf16 parameter casts of a network in one launch group (train_step.py)",
    "gn_resident": "one-launch GroupNorm keeps small strips in registers: one memory round trip (ops.py, csrc/norm.hip)",
    "head_major_qk": "the chain kernels write q / k of the d = 40 attention as [sample][head][token][40] images (fused.py)",
    # ---- on-switches of measured-slower paths ----
    "splitk_gn": "GroupNorm as the split-K second pass (0.07 ms slower: profiles/r04_splitk_gn_ab.txt)",
    "colsum_one_launch": "column sums with an in-launch ticket instead of two launches (slower)",
    "fused_colsum": "bias gradient sums inside the dx GEMM launch (slower)",
    "side_stream": "=1|2|3: independent launches on a sibling graph branch (0.5 ms slower: DESIGN.md)",
    "wsconv": "weight-streaming conv kernel where K >= wsconv_min_k (needs `make WSCONV=1`; slower in the step)",
    # ---- numeric knobs of the A/B runs ----
    "wsconv_min_k": "", "wsconv_waves": "", "zero_page_bytes": "", "conv_cblock": "", "gn_stat_kb": "", "gn_apply_kb": "",
    "gn_apply_max": "", "gn_fused_max_rows": "", "gn_bwd_fused_max_rows": "", "wgrad_tile": "", "wgrad_splits": "",
    "flash_direct_min_d": "", "tchain_min_rows": "", "gn_resident_max_rows": "",
}

# Environment variables of rounds 1-4 that UR_EXPERIMENT replaced.  They are NOT read any more; a script that still sets one
# would silently measure the default path, so their presence is an error.  UR_NORM_XCD / UR_FLASH_M32 / UR_NO_*
# style C++ toggles were hardwired in the same prune.
RETIRED = {
    "UR_ATTN_BWD_TRN", "UR_BATCH_CASTS", "UR_BATCH_WT", "UR_COLSUM_ONE_LAUNCH", "UR_CONV_CBLOCK", "UR_CTX3_EARLY",
    "UR_CTXKV_ONE_LAUNCH", "UR_EXCHANGE_EARLY", "UR_FLASH_BACKWARD", "UR_FLASH_DIRECT_MIN_D", "UR_FLASH_M32", "UR_FOLD_SHORTCUT",
    "UR_FORWARD_LSE", "UR_FUSED_COLSUM", "UR_FUSED_GRADNORM", "UR_GNF_XCD", "UR_GN_APPLY_KB", "UR_GN_APPLY_MAX",
    "UR_GN_BWD_FUSED_MAX_ROWS", "UR_GN_FUSED_MAX_ROWS", "UR_GN_STAT_KB", "UR_HEADS_MULTI", "UR_MULTI_TRANSPOSE", "UR_NORM_DEFER",
    "UR_NORM_XCD", "UR_QKV_ONE_LAUNCH", "UR_SIDE_STREAM", "UR_SPLITK_GN", "UR_TCHAIN", "UR_TCHAIN_FF", "UR_TCHAIN_PRE",
    "UR_TCHAIN_Q", "UR_TRANSPOSE_MAX", "UR_VT_FIRST", "UR_WGRAD", "UR_WGRAD_DEFER", "UR_WGRAD_SPLITS", "UR_WGRAD_TABLE",
    "UR_WGRAD_TILE", "UR_WSCONV", "UR_WSCONV_MIN_K", "UR_WSCONV_WAVES", "UR_ZERO_PAGE_BYTES",
}

_parsed = None


def _load():
    global _parsed
    if _parsed is None:
        stale = sorted(k for k in os.environ if k in RETIRED)
        if stale:
            raise ValueError(f"retired environment variable(s) {', '.join(stale)}: no longer read -- use UR_EXPERIMENT=... "
                             f"(uni_renderer_amd/_experiments.py lists the entries)")
        d = {}
        for item in os.environ.get("UR_EXPERIMENT", "").replace(";", ",").split(","):
            item = item.strip()
            if not item:
                continue
            name, _, val = item.partition("=")
            neg = name.startswith("no_") and name[3:] in KNOWN
            key = name[3:] if neg else name
            if key not in KNOWN:
                raise ValueError(f"UR_EXPERIMENT: unknown entry {item!r} (known: {', '.join(sorted(KNOWN))})")
            d[key] = "0" if neg else (val if val else "1")
        _parsed = d
    return _parsed


def flag(name: str, default: bool) -> bool:
    assert name in KNOWN, name
    v = _load().get(name)
    return default if v is None else v != "0"


def number(name: str, default: int) -> int:
    assert name in KNOWN, name
    v = _load().get(name)
    return default if v is None else int(v)
