#!/usr/bin/env python3
"""Generates uni_renderer_amd/csrc/tchain_asm.inc: the hand-scheduled ds_read / MFMA streams of csrc/tchain.hip.

Why inline asm: with ONE wave per SIMD nothing hides the LDS latency of an A-fragment read except issuing it several
MFMAs early, and hipcc (ROCm 7.2) sinks every ds_read_b128 next to its consumer when the kernel is near the register
limit -- with or without sched_group_barrier -- which measured 180-240 cycles per MFMA instead of 32.  The strings below
fix the order: AHEAD fragment reads in flight, `s_waitcnt lgkmcnt(n)` counted (LDS returns in order), one read issued
behind every MFMA.  Operand names: c* accumulators (AGPR tuples), f* fragment ring (VGPR), a0..a3 the lane's fragment
address for k16 step s of the stage (slot base + row * 128 + swizzled chunk), b* B operands (VGPR).

    python tools/gen_tchain_asm.py        # rewrites the .inc next to tchain.hip
"""
import os

RD, AHEAD = 12, 10


NPIECE = 10  # LDS-DMA pieces (1 KiB each) a wave copies per stage


def dma_ops():
    """The wave's ten `buffer_load ... lds` of the stage two ahead, as (instruction list) per piece: M0 = LDS destination,
    soffset = byte offset into the weight stream; both advance by 1 KiB per piece (operands dm0 / dso are scratch SGPRs
    initialised from ld0 / so0)."""
    ops = []
    for i in range(NPIECE):
        pre = ["s_mov_b32 %[dso], %[so0]", "s_mov_b32 m0, %[ld0]"] if i == 0 else ["s_add_u32 %[dso], %[dso], 0x400", "s_add_u32 m0, m0, 0x400"]
        ops.append(pre + ["s_nop 0", "buffer_load_dwordx4 %[vo], %[rs], %[dso] offen lds"])
    return ops


def operand_dma_ops(n):
    """LDS-DMA pieces of the NEXT operand block of csrc/wsconv.hip (n per stage): a `buffer_load ... lds` through the
    source tensor's descriptor %[ors] with the per-lane byte offsets %[q<j>] (bit 31 set = zero padding: out of range
    reads deliver 0), wave-uniform tap / channel-block offset %[oso], LDS destination %[ob] + j KiB."""
    ops = []
    for j in range(n):
        pre = ["s_mov_b32 m0, %[ob]"] if j == 0 else [f"s_add_u32 m0, %[ob], 0x{j * 0x400:x}"]
        ops.append(pre + ["s_nop 0", f"buffer_load_dwordx4 %[q{j}], %[ors], %[oso] offen lds"])
    return ops


def stream(n, acc_of, frag_addr, b_of, valu=(), dma=False, c0=lambda m: None, opieces=0, wpieces=NPIECE):
    """n MFMAs; MFMA m multiplies fragment m (read from frag_addr(m) = (address operand, immediate offset)) with B operand
    b_of(m) into accumulator acc_of(m) (c0(m): literal C operand of a first MFMA).  `valu`: a VALU program spread evenly
    behind the MFMAs; `dma`: the LDS-DMA pieces of the next-but-one stage, one behind every fourth MFMA."""
    out = ["s_nop 1", "s_waitcnt lgkmcnt(0)"]

    def rd(m):
        a, off = frag_addr(m)
        return f"ds_read_b128 %[f{m % RD}], %[{a}] offset:{off}"

    if n == 0:
        return out + list(valu)
    for m in range(min(AHEAD, n)):
        out.append(rd(m))
    pieces = dma_ops()[:wpieces] if dma else []
    where = {4 * i + 1: p for i, p in enumerate(pieces)}
    if n == 20:  # half-width conv stage (8-wave workgroup): 5 weight pieces behind every third MFMA, then the operand pieces
        where = {3 * i + 1: p for i, p in enumerate(pieces)}
        where.update({3 * len(pieces) + 1 + j: p for j, p in enumerate(operand_dma_ops(opieces))})
    elif opieces:  # weight pieces first, the operand pieces last: `s_waitcnt vmcnt(opieces)` then guarantees the
        # weights of the next stage while the operand pieces may still be in flight.  EARLY: the ring is only two slots
        # deep (the rest of the LDS stages operands), so the copies of stage g + 1 have just the rest of stage g to land
        # (issue -> landed is ~1.1 us on a loaded chip): one behind each of the first ten MFMAs, not spread over thirty.
        step = int(os.environ.get("UR_GEN_CONV_WSTEP", "3"))
        where = {step * i + 1: p for i, p in enumerate(pieces)}
        o0 = step * (len(pieces) - 1) + 3
        where.update({o0 + 2 * j: p for j, p in enumerate(operand_dma_ops(opieces))})
    per = -(-len(valu) // (n - 1)) if valu else 0
    pi = 0
    for m in range(n):
        out.append(f"s_waitcnt lgkmcnt({min(AHEAD - 1, n - 1 - m)})")
        c = c0(m) or f"%[{acc_of(m)}]"
        out.append(f"v_mfma_f32_32x32x16_\" MT \" %[{acc_of(m)}], %[f{m % RD}], %[{b_of(m)}], {c}")
        if m + AHEAD < n:
            out.append(rd(m + AHEAD))
        if m in where:
            out += where[m]
        if m >= 1 and valu:
            out += valu[pi: pi + per]
            pi += per
    out += valu[pi:]
    return out


def cstr(lines):
    return " \\\n".join(f'    "{l}\\n\\t"' for l in lines)


import math

K_Z = math.sqrt(0.5 * 1.4426950408889634)   # zc = |g| * K_Z:  zc^2 = (g^2 / 2) * log2(e)
GELU = dict(kz=K_Z, p=0.3275911 * 0.70710678118654752 / K_Z, a5=1.061405429, a4=-1.453152027, a3=1.421413741,
            a2=-0.284496736, a1=0.254829592)


def geglu_program():
    """VALU program of one GEGLU half-chunk (32 hidden units = one accumulator tile each of value / gate rows, in VGPRs):
         h2[i] = (pv[i] + bv[i]) * (g + |g| * erf(|g| / sqrt 2)),  g = pg[i] + bg[i]          (= 2 * value * gelu(gate); the
       0.5 is folded into the second feed-forward matrix by the host), erf by Abramowitz-Stegun 7.1.26 as gelu_erf_f in
       ur_common.h, with zc = |g| sqrt(log2(e) / 2) so that exp(-z^2) = exp2(-zc^2) and 1 + p z = 1 + p' zc.  Four values
       (r = 0..3 of one 8-row block q) run interleaved so that a dependent instruction is four issues away; results are
       packed in pairs into o[(q >> 1) * 4 + 2 * (q & 1) + (r >> 1)]: the B operands of k16 steps 0 / 1 of the half-chunk in
       the KPERM order."""
    prog = []
    for q in range(4):
        steps = [
            lambda r, i: f"v_add_f32 %[g{r}], %[pg{i}], %[bg{i}]",
            lambda r, i: f"v_mul_f32_e64 %[e{r}], |%[g{r}]|, %[ks0]",
            lambda r, i: f"v_fma_f32 %[t{r}], %[e{r}], %[ks1], 1.0",
            lambda r, i: f"v_rcp_f32 %[t{r}], %[t{r}]",
            lambda r, i: f"v_mul_f32_e64 %[e{r}], -%[e{r}], %[e{r}]",
            lambda r, i: f"v_exp_f32 %[e{r}], %[e{r}]",
            lambda r, i: f"v_fma_f32 %[p{r}], %[t{r}], %[ks2], %[kv4]",
            lambda r, i: f"v_fma_f32 %[p{r}], %[p{r}], %[t{r}], %[kv3]",
            lambda r, i: f"v_fma_f32 %[p{r}], %[p{r}], %[t{r}], %[kv2]",
            lambda r, i: f"v_fma_f32 %[p{r}], %[p{r}], %[t{r}], %[kv1]",
            lambda r, i: f"v_mul_f32 %[p{r}], %[p{r}], %[t{r}]",
            lambda r, i: f"v_fma_f32 %[p{r}], -%[p{r}], %[e{r}], 1.0",
            lambda r, i: f"v_mul_f32_e64 %[p{r}], |%[g{r}]|, %[p{r}]",
            lambda r, i: f"v_add_f32 %[g{r}], %[g{r}], %[p{r}]",
            lambda r, i: f"v_add_f32 %[e{r}], %[pv{i}], %[bv{i}]",
            lambda r, i: f"v_mul_f32 %[g{r}], %[e{r}], %[g{r}]",
        ]
        for st in steps:
            for r in range(4):
                prog.append(st(r, 4 * q + r))
        o = (q >> 1) * 4 + 2 * (q & 1)
        prog.append(f"v_cvt_pk_\" MT \"_f32 %[o{o}], %[g0], %[g1]")
        prog.append(f"v_cvt_pk_\" MT \"_f32 %[o{o + 1}], %[g2], %[g3]")
    return prog


def ffa_stream(with_mfma, with_g, dma=True):
    """Feed-forward input stage: 20 (sub-image c, step s) pairs i, MFMA 2 i = value rows into hv, 2 i + 1 = gate rows into
    hg (the first MFMA of each starts from C = 0), interleaved with the GEGLU program of the PREVIOUS half-chunk and the
    LDS-DMA pieces of the stage two ahead."""
    return stream(40 if with_mfma else 0, lambda m: ("hv", "hg")[m & 1],
                  lambda m: (f"a{(m >> 1) & 3}", (m >> 3) * 8192 + (m & 1) * 4096), lambda m: f"b{m >> 1}",
                  valu=geglu_program() if with_g else (), dma=dma and with_mfma, c0=lambda m: "0" if m < 2 else None)


DMA_OUTS = ['[dso] "=&s"(t_dso)']
DMA_INS = ['[vo] "v"(DVO)', '[rs] "s"(DRS)', '[so0] "s"(DSO)', '[ld0] "s"(DLD)']


def operands_ffag():
    outs = ['[hv] "=&v"(HV)', '[hg] "=&v"(HG)'] + [f'[o{k}] "=&v"(O[{k}])' for k in range(8)]
    outs += [f'[f{k}] "=&v"(f{k})' for k in range(RD)]
    outs += [f'[{n}{r}] "=&v"(t_{n}{r})' for n in "getp" for r in range(4)]
    ins = [f'[a{k}] "v"(A[{k}])' for k in range(4)] + [f'[b{k}] "a"(B[{k}])' for k in range(20)]
    ins += [f'[pv{i}] "v"(PV[{i}])' for i in range(16)] + [f'[pg{i}] "v"(PG[{i}])' for i in range(16)]
    ins += [f'[bv{i}] "v"(BV[{i >> 2}][{i & 3}])' for i in range(16)] + [f'[bg{i}] "v"(BG[{i >> 2}][{i & 3}])' for i in range(16)]
    ins += [f'[ks{k}] "s"(KS[{k}])' for k in range(3)] + [f'[kv{k}] "v"(KV[{k}])' for k in range(1, 5)]
    return outs, ins


def main():
    import re
    # N = 320 GEMM stage: m = 10 s + t: tile t (32 rows at t * 4096), k16 step s
    gargs = (40, lambda m: f"c{m % 10}", lambda m: (f"a{m // 10}", (m % 10) * 4096), lambda m: f"b{m // 10}")
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "uni_renderer_amd", "csrc", "tchain_asm.inc")
    outs, ins = operands_ffag()
    is_mfma_side = lambda x: re.match(r"\[(hv|hg|f\d+|a\d+|b\d+)\]", x) is not None
    g_outs = [o for o in outs if not is_mfma_side(o)]
    g_ins = [i for i in ins if not is_mfma_side(i)]
    a_outs = [o for o in outs if is_mfma_side(o)]
    a_ins = [i for i in ins if is_mfma_side(i)]
    fmt = lambda xs: ", ".join(xs)
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_tchain_asm.py -- do not edit.  MT = \"f16\" or \"bf16\" (string literal).\n")
        f.write(f"// fragment ring {RD}, {AHEAD} reads in flight; GEGLU constants: {GELU}\n")
        f.write(f"#define TC_GELU_KS {{{GELU['kz']!r}f, {GELU['p']!r}f, {GELU['a5']!r}f}}\n")
        f.write(f"#define TC_GELU_KV {{0.f, {GELU['a1']!r}f, {GELU['a2']!r}f, {GELU['a3']!r}f, {GELU['a4']!r}f}}\n\n")
        f.write("// N = 320 GEMM stage, without / with the LDS-DMA pieces of the stage two ahead\n")
        f.write("#define TC_ASM_GEMM_STAGE(MT) \\\n" + cstr(stream(*gargs)) + "\n\n")
        f.write("#define TC_ASM_GEMM_STAGE_DMA(MT) \\\n" + cstr(stream(*gargs, dma=True)) + "\n\n")
        f.write("// conv stage (csrc/wsconv.hip): weights of the next stage + 5 pieces of the next operand block\n")
        f.write("#define TC_ASM_CONV_STAGE(MT) \\\n" + cstr(stream(*gargs, dma=True, opieces=5)) + "\n\n")
        f.write("// half-width conv stage (8 waves per workgroup, 2 per SIMD: a wave owns 32 pixels x 160 channels = tiles c0..c4):\n")
        f.write("// 20 MFMAs, 5 weight pieces, 0 / 3 / 4 operand pieces\n")
        hargs = (20, lambda m: f"c{m % 5}", lambda m: (f"a{m // 5}", (m % 5) * 4096), lambda m: f"b{m // 5}")
        for npc in (0, 3, 4):
            f.write(f"#define TC_ASM_CONV8_STAGE_O{npc}(MT) \\\n" + cstr(stream(*hargs, dma=True, opieces=npc, wpieces=5)) + "\n\n")
        f.write("// feed-forward input stage + the GEGLU program of the previous half-chunk + LDS-DMA (tools/gen_tchain_asm.py)\n")
        f.write("#define TC_ASM_FFAG(MT) \\\n" + cstr(ffa_stream(True, True)) + "\n\n")
        f.write("#define TC_ASM_FFA(MT) \\\n" + cstr(ffa_stream(True, False)) + "\n\n")
        f.write("#define TC_ASM_G(MT) \\\n" + cstr(ffa_stream(False, True)) + "\n\n")
        f.write("// operand lists: HV / HG accumulators written (VGPR tuples), PV / PG the previous pair (read element-wise), O[8]\n")
        f.write("// packed output words, A[4] fragment addresses, B[20] operands (AGPR), BV / BG [4] float4 biases, KS / KV GELU\n")
        f.write("// constants, DVO / DRS / DSO / DLD the LDS-DMA lane offset, buffer descriptor, stream offset, LDS destination\n")
        f.write(f"#define TC_OPS_FFAG(HV, HG, PV, PG, O, A, B, BV, BG, KS, KV, DVO, DRS, DSO, DLD) : {fmt(outs + DMA_OUTS)} : {fmt(ins + DMA_INS)} : \"memory\"\n")
        f.write(f"#define TC_OPS_FFA(HV, HG, A, B, DVO, DRS, DSO, DLD) : {fmt(a_outs + DMA_OUTS)} : {fmt(a_ins + DMA_INS)} : \"memory\"\n")
        f.write(f"#define TC_OPS_G(PV, PG, O, BV, BG, KS, KV) : {fmt(g_outs)} : {fmt(g_ins)}\n")
        f.write(f"#define TC_OPS_DMA(DVO, DRS, DSO, DLD) {fmt(DMA_OUTS)}, {fmt(DMA_INS)}\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
