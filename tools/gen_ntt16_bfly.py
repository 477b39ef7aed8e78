#!/usr/bin/env python3
"""Generates nested_hashing_psi_amd/csrc/ntt16_bfly.inc: the 60-bit lazy NTT butterflies of ntt16_kernel.h as hand-scheduled
gfx950 instruction blocks (the generator can interleave NB independent butterflies per block; NB = 1 is what ships).

Issue costs on gfx950 (tools/microbench_ops.hip, cycles per wave64 instruction per SIMD with 2+ waves): every VOP3 instruction --
v_mad_u64_u32, v_lshl_add_u64, v_lshrrev_b64, v_bfi_b32, v_mul_lo/hi_u32 alike -- 4.1-4.3; 32-bit VOP1/VOP2 2.0-2.5; a wave
alone on its SIMD 4+ for everything; a 2-cycle instruction that follows a 4-cycle one costs 4 itself (tools/microbench_operands.hip), so
the VOP1/VOP2 instructions are grouped.  Forward block 21 instructions, inverse 22 (its last product is written straight to the output).

    python tools/gen_ntt16_bfly.py          (rewrites the .inc; the output is committed)

Register use of stream i: n = 11 (forward) or 13 (inverse) fixed VGPRs v[128 - n (i + 1) .. 128 - n i) (an asm operand cannot name the halves of a
64-bit pair, so every temporary whose halves are needed lives in a named register; all are in the clobber list).
"""
import os

OUT = os.environ.get("NTT16_BFLY_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nested_hashing_psi_amd", "csrc", "ntt16_bfly.inc")


NREG = {False: 11, True: 13}  # fixed VGPRs per stream: forward (CT), inverse (GS)


class Regs:
    """Fixed registers of stream i.  Lifetimes allow these overlays (see the instruction lists):
    both: U over T (the select / t writes in place); the sign mask in QE's low register until qe is written
    CT:   N over CR (cr is dead once qe exists), T6 over QE (computed after the last use of qe)
    GS:   N over CR (~b is dead before m1), the early a + 4q + 1 over QE, the masked 4q over D (d is dead by then)"""
    def __init__(self, i, gs):
        n = NREG[gs]
        self.n = n
        self.base = 128 - n * (i + 1) - int(os.environ.get('NTT16_BASE_SHIFT', '0'))
        b = self.base
        self.X = b           # 2 bh
        self.M = b           # sign mask
        self.T = b + 1       # t = x - 4q
        self.U = self.T      # CT: u (in place); GS: s, then t in place
        self.CR = b + 3      # m1 / cr / cr >> 31
        self.N = self.CR     # ~v (CT, after qe) / ~b (GS, before m1)
        self.QE = b + 5
        self.T6 = self.QE    # CT: u + 4q + 1 (after the last use of qe); GS: a + 4q + 1 (before qe)
        self.ACC = b + 7     # acc, then v
        self.C = b + 9
        self.D = b + 11      # GS only: d = a - b + 4q

    def all(self):
        return range(self.base, self.base + self.n)


def p(r):
    return "v[%d:%d]" % (r, r + 1)


def v(r):
    return "v%d" % r


def ct_stream(i, h2=False):
    r = Regs(i, False)
    o = lambda name: "%%[%s%d]" % (name, i)
    bl, bh = o("bl"), o("bh")
    m = r.QE  # the sign mask lives in qe's low register until qe is written (the selects consume it before that)
    return [
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.CR), bl, o("sh")),                  # m1 = bl sh
        "v_lshl_add_u64 %s, %s, 0, %%[nq4]" % (p(r.T), o("a")),                       # t = a - 4q
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.CR), bh, o("sl"), p(r.CR)),        # cr = bh sl + m1
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.ACC), bl, o("wl")),                 # acc = bl wl
        # the 2-cycle instructions sit next to each other: one that follows a 4-cycle instruction costs 4 itself
        # (profiles/r03/microbench_operands.txt, "mad + v_not alternating")
    ] + ([] if h2 else [
        "v_lshlrev_b32 %s, 1, %s" % (v(r.X), bh),                                     # 2 bh
    ]) + [
        "v_ashrrev_i32 %s, 31, %s" % (v(m), v(r.T + 1)),                              # all ones iff t < 0
        "v_bfi_b32 %s, %s, %s, %s" % (v(r.U), v(m), o("al"), v(r.T)),                 # u = t < 0 ? a : t   (in place)
        "v_bfi_b32 %s, %s, %s, %s" % (v(r.U + 1), v(m), o("ah"), v(r.T + 1)),
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.C), bl, o("wh")),                   # c = bl wh
        "v_lshrrev_b64 %s, 31, %s" % (p(r.CR), p(r.CR)),
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.C), bh, o("wl"), p(r.C)),          # c += bh wl
        ("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.QE), bh, o("sh2"), p(r.CR))) if h2 else     # qe = bh (2 sh) + (cr >> 31)
        ("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.QE), v(r.X), o("sh"), p(r.CR))),  # qe = 2 bh sh + (cr >> 31)
        "v_mad_u64_u32 %s, vcc, %s, %%[nql], %s" % (p(r.ACC), v(r.QE), p(r.ACC)),     # acc += qe_lo nq_lo
        "v_mad_u64_u32 %s, vcc, %s, %%[nqh], %s" % (p(r.C), v(r.QE), p(r.C)),         # c += qe_lo nq_hi
        "v_mad_u64_u32 %s, vcc, %s, %%[nql], %s" % (p(r.C), v(r.QE + 1), p(r.C)),     # c += qe_hi nq_lo
        "v_lshl_add_u64 %s, %s, 0, %%[q4p1]" % (p(r.T6), p(r.U)),                     # u + 4q + 1   (over qe)
        "v_not_b32 %s, %s" % (v(r.N), v(r.ACC)),
        "v_add_u32 %s, %s, %s" % (v(r.ACC + 1), v(r.ACC + 1), v(r.C)),                # v = acc + (c << 32)
        "v_not_b32 %s, %s" % (v(r.N + 1), v(r.ACC + 1)),
        "v_lshl_add_u64 %s, %s, 0, %s" % (o("ao"), p(r.U), p(r.ACC)),                 # a' = u + v
        "v_lshl_add_u64 %s, %s, 0, %s" % (o("bo"), p(r.T6), p(r.N)),                  # b' = u + 4q + 1 + ~v
    ]


def gs_stream(i, h2=False):
    """h2: wave-uniform twiddles only (SGPR operands), with 2 sh as an operand of its own.  No per-lane input is read after the
    first six instructions then, so a' may be written early, d's registers stay intact for the quotient estimate, and the
    doubling of dh goes away: 21 instructions."""
    r = Regs(i, True)
    o = lambda name: "%%[%s%d]" % (name, i)
    dl, dh = v(r.D), v(r.D + 1)
    head = [
        "v_not_b32 %s, %s" % (v(r.N), o("bl")),
        "v_not_b32 %s, %s" % (v(r.N + 1), o("bh")),
        "v_lshl_add_u64 %s, %s, 0, %%[q4p1]" % (p(r.T6), o("a")),                     # a + 4q + 1   (over qe, dead here)
        "v_lshl_add_u64 %s, %s, 0, %s" % (p(r.U), o("a"), o("b")),                    # s = a + b
        "v_lshl_add_u64 %s, %s, 0, %s" % (p(r.D), p(r.T6), p(r.N)),                   # d = a - b + 4q
        "v_lshl_add_u64 %s, %s, 0, %%[nq4]" % (p(r.T), p(r.U)),                       # t = s - 4q   (in place)
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.ACC), dl, o("wl")),
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.C), dl, o("wh")),
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.CR), dl, o("sh")),                  # (over ~b, dead)
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.CR), dh, o("sl"), p(r.CR)),
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.C), dh, o("wl"), p(r.C)),
    ]
    tail = [
        "v_mad_u64_u32 %s, vcc, %s, %%[nqh], %s" % (p(r.C), v(r.QE), p(r.C)),
        "v_mad_u64_u32 %s, vcc, %s, %%[nql], %s" % (p(r.C), v(r.QE + 1), p(r.C)),
        "v_add_u32 %s, %s, %s" % (v(r.ACC + 1), v(r.ACC + 1), v(r.C)),                # dl wl + (c << 32)
    ]
    last = "v_mad_u64_u32 %s, vcc, %s, %%[nql], %s" % (o("bo"), v(r.QE), p(r.ACC))   # b' = d w: the last product lands in the output
    if h2:
        m = r.X
        return head + [
            "v_ashrrev_i32 %s, 31, %s" % (v(m), v(r.T + 1)),
            "v_and_b32 %s, %%[q4l], %s" % (v(r.QE), v(m)),                            # 4q where t < 0   (qe's registers, until qe exists)
            "v_and_b32 %s, %%[q4h], %s" % (v(r.QE + 1), v(m)),
            "v_lshl_add_u64 %s, %s, 0, %s" % (o("ao"), p(r.T), p(r.QE)),              # a' = t < 0 ? s : t
            "v_lshrrev_b64 %s, 31, %s" % (p(r.CR), p(r.CR)),
            "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.QE), dh, o("sh2"), p(r.CR)),
        ] + tail + [last]
    m = r.T6  # sign mask: in the register of a + 4q + 1 (dead once d exists), until qe is written
    return head + [
        # four 2-cycle instructions in a row (see ct_stream); d is dead after the first of them and takes the masked 4q
        "v_lshlrev_b32 %s, 1, %s" % (v(r.X), dh),
        "v_ashrrev_i32 %s, 31, %s" % (v(m), v(r.T + 1)),
        "v_and_b32 %s, %%[q4l], %s" % (dl, v(m)),                                     # 4q where t < 0
        "v_and_b32 %s, %%[q4h], %s" % (dh, v(m)),
        "v_lshrrev_b64 %s, 31, %s" % (p(r.CR), p(r.CR)),
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.QE), v(r.X), o("sh"), p(r.CR)),    # the last read of a per-lane twiddle
    ] + tail + [
        "v_lshl_add_u64 %s, %s, 0, %s" % (o("ao"), p(r.T), p(r.D)),                   # a' = t < 0 ? s : t
        last,
    ]


def interleave(streams):
    out = []
    for k in range(max(len(s) for s in streams)):
        for s in streams:
            if k < len(s):
                out.append(s[k])
    return out


def emit(name, gs, nb, h2=False):
    """h2: the twiddle's doubled high Shoup word is an operand (t.sh2); the inverse form takes wave-uniform twiddles only"""
    lines = interleave([(gs_stream if gs else ct_stream)(i, h2) for i in range(nb)])
    body = " \\\n".join('        "%s\\n\\t"' % ln for ln in lines)
    # no early-clobber: every per-lane input is read before the first output is written (checked below), so outputs may reuse
    # input registers
    first_out = min(k for k, ln in enumerate(lines) if "%[ao" in ln or "%[bo" in ln)
    tw_names = () if (gs and h2) else ("wl", "wh", "sl", "sh", "sh2")   # (scalar registers in the uniform-only form)
    vgpr_inputs = ["%%[%s%d]" % (nm, i) for nm in ("a", "b", "al", "ah", "bl", "bh") + tw_names for i in range(nb)]
    for ln in lines[first_out + 1:]:
        assert not any(op + "," in ln + "," or ln.endswith(op) for op in vgpr_inputs), "input read after an output was written: " + ln
    outs = ", ".join('[ao%d] "=v"(ao%d), [bo%d] "=v"(bo%d)' % (i, i, i, i) for i in range(nb))
    twc = "NTT16_S" if (gs and h2) else "TWC"
    ins = []
    for i in range(nb):
        if gs:
            ins.append('[a%d] "v"(a%d), [b%d] "v"(b%d), [bl%d] "v"((u32)b%d), [bh%d] "v"((u32)(b%d >> 32))' % ((i,) * 8))
        else:
            ins.append('[a%d] "v"(a%d), [al%d] "v"((u32)a%d), [ah%d] "v"((u32)(a%d >> 32)), [bl%d] "v"((u32)b%d), [bh%d] "v"((u32)(b%d >> 32))'
                       % ((i,) * 10))
        ins.append('[wl%d] %s(t%d.wl), [wh%d] %s(t%d.wh), [sl%d] %s(t%d.sl), [sh%d] %s(t%d.sh)' % (i, twc, i, i, twc, i, i, twc, i, i, twc, i))
        if h2:
            ins.append('[sh2%d] %s(t%d.sh2)' % (i, twc, i))
    ins.append('[nql] "s"(m.nql), [nqh] "s"(m.nqh), [nq4] "s"(m.nq4), [q4p1] "s"(m.q4p1)')
    if gs:
        ins.append('[q4l] "s"((u32)(m.q4p1 - 1)), [q4h] "s"((u32)((m.q4p1 - 1) >> 32))')
    clob = ['"vcc"'] + ['"v%d"' % x for i in range(nb) for x in Regs(i, gs).all()]
    args = "" if (gs and h2) else "TWC"
    return ("#define %s(%s) \\\n    asm( \\\n%s \\\n        : %s \\\n        : %s \\\n        : %s)\n"
            % (name, args, body, outs, ", \\\n          ".join(ins), ", ".join(clob)))


def main():
    text = ("// ntt16_bfly.inc -- GENERATED by tools/gen_ntt16_bfly.py; do not edit.  See that script and ntt16_kernel.h.\n"
            "// NTT16_{CT,GS}1(TWC): one butterfly on (a0, b0, t0) -> (ao0, bo0);\n"
            "// TWC = NTT16_S (wave-uniform twiddles, SGPR operands) or NTT16_V (per-lane twiddles); m = ModC.\n"
            "// NTT16_CT1H(TWC) / NTT16_GS1HS(): 2 sh is an operand of its own (t0.sh2) -- one instruction fewer; the inverse form is\n"
            "// for wave-uniform twiddles only.\n")
    # one butterfly per block: measured, a wave issues at most every other VALU slot whatever its instruction-level parallelism
    # (interleaving two butterflies per block changed nothing but the register count), so parallelism comes from waves
    text += emit("NTT16_CT1", False, 1) + emit("NTT16_GS1", True, 1) + emit("NTT16_CT1H", False, 1, True) + emit("NTT16_GS1HS", True, 1, True)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
