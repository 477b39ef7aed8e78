#!/usr/bin/env python3
"""Generates nested_hashing_psi_amd/csrc/ntt16_bfly.inc: the 60-bit lazy NTT butterflies of ntt16_kernel.h as hand-scheduled
gfx950 instruction blocks (the generator can interleave NB independent butterflies per block; NB = 1 is what ships).

Issue costs on gfx950 (tools/microbench_ops.hip, cycles per wave64 instruction per SIMD with 2+ waves): every VOP3 instruction --
v_mad_u64_u32, v_lshl_add_u64, v_lshrrev_b64, v_bfi_b32, v_mul_lo/hi_u32 alike -- 4.1-4.3; 32-bit VOP1/VOP2 2.0-2.5; a wave
alone on its SIMD 4+ for everything; a 2-cycle instruction that follows a 4-cycle one costs 4 itself (tools/microbench_operands.hip), so
the VOP1/VOP2 instructions are grouped.  Both blocks are 20 instructions, 19 where 2 sh is an operand (the inverse's last product is
written straight to the output; 64-bit differences are borrow chains through VCC, see ct_stream).

    python tools/gen_ntt16_bfly.py          (rewrites the .inc; the output is committed)

Register use of stream i: 11 fixed VGPRs v[128 - 11 (i + 1) .. 128 - 11 i) (an asm operand cannot name the halves of a 64-bit pair, so every
temporary whose halves are needed lives in a named register; all are in the clobber list).
"""
import os

OUT = os.environ.get("NTT16_BFLY_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nested_hashing_psi_amd", "csrc", "ntt16_bfly.inc")


NREG = {False: 11, True: 11}  # fixed VGPRs per stream: forward (CT), inverse (GS)


class Regs:
    """Fixed registers of stream i.  Lifetimes allow these overlays (see the instruction lists):
    CT:   u over t (the select writes in place); the sign mask in QE's low register until qe is written; u + 4q over QE (after
          the last use of qe)
    GS:   a + 4q over QE (dead once d exists); s = a + b over ACC and t = s - 4q over C (both dead after the select, before
          the first product); the sign mask in X until 2 dh is written"""
    def __init__(self, i, gs):
        n = NREG[gs]
        self.n = n
        self.base = 128 - n * (i + 1) - int(os.environ.get('NTT16_BASE_SHIFT', '0'))
        b = self.base
        self.X = b           # 2 bh (CT) / sign mask, then 2 dh (GS)
        self.T = b + 1       # CT: t = a - 4q, then u in place
        self.U = self.T
        self.D = b + 1       # GS: d = a - b + 4q
        self.CR = b + 3      # m1 / cr / cr >> 31
        self.QE = b + 5
        self.T6 = self.QE    # CT: u + 4q (after the last use of qe); GS: a + 4q (before qe)
        self.ACC = b + 7     # acc, then v
        self.C = b + 9

    def all(self):
        return range(self.base, self.base + self.n)


def p(r):
    return "v[%d:%d]" % (r, r + 1)


def v(r):
    return "v%d" % r


# A 64-bit difference is v_sub_co_u32 + v_subb_co_u32 through VCC (gfx950 has no 64-bit vector subtraction; the alternative,
# x + ~y + 1, is two v_not_b32 and a v_lshl_add_u64).  A VALU instruction that reads VCC needs two wait states after the VALU
# write of VCC: two independent instructions that leave VCC alone sit between the halves (the multiplier instructions discard
# their carry-out into VCC, so none of them may).  The halves of an asm operand cannot be named, so a result that is produced
# in halves is two 32-bit outputs (the register coalescer makes them a pair).
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
        "v_mad_u64_u32 %s, vcc, %s, %%[nql], %s" % (p(r.C), v(r.QE + 1), p(r.C)),     # c += qe_hi nq_lo: the last read of an input
        "v_lshl_add_u64 %s, %s, 0, %%[q4]" % (p(r.T6), p(r.U)),                       # u + 4q   (over qe)
        "v_sub_co_u32 %s, vcc, %s, %s" % (o("bol"), v(r.T6), v(r.ACC)),               # b' = u + 4q - v, low half
        "v_add_u32 %s, %s, %s" % (v(r.ACC + 1), v(r.ACC + 1), v(r.C)),                # v = acc + (c << 32)
        "v_lshl_add_u64 %s, %s, 0, %s" % (o("ao"), p(r.U), p(r.ACC)),                 # a' = u + v
        "v_subb_co_u32 %s, vcc, %s, %s, vcc" % (o("boh"), v(r.T6 + 1), v(r.ACC + 1)), # b', high half
    ]


def gs_stream(i, h2=False):
    """h2: 2 sh is an operand of its own (wave-uniform twiddles: the doubling is a scalar instruction)"""
    r = Regs(i, True)
    o = lambda name: "%%[%s%d]" % (name, i)
    dl, dh = v(r.D), v(r.D + 1)
    s, t, m = r.ACC, r.C, r.X
    return [
        "v_lshl_add_u64 %s, %s, 0, %%[q4]" % (p(r.T6), o("a")),                       # a + 4q   (over qe)
        "v_lshl_add_u64 %s, %s, 0, %s" % (p(s), o("a"), o("b")),                      # s = a + b   (over acc)
        "v_sub_co_u32 %s, vcc, %s, %s" % (dl, v(r.T6), o("bl")),                      # d = a + 4q - b, low half
        "v_lshl_add_u64 %s, %s, 0, %%[nq4]" % (p(t), p(s)),                           # t = s - 4q   (over c)
        "v_ashrrev_i32 %s, 31, %s" % (v(m), v(t + 1)),                                # all ones iff t < 0
        "v_subb_co_u32 %s, vcc, %s, %s, vcc" % (dh, v(r.T6 + 1), o("bh")),            # d, high half
        "v_bfi_b32 %s, %s, %s, %s" % (o("aol"), v(m), v(s), v(t)),                    # a' = t < 0 ? s : t   (early-clobber outputs:
        "v_bfi_b32 %s, %s, %s, %s" % (o("aoh"), v(m), v(s + 1), v(t + 1)),            #  the twiddles are read below)
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.ACC), dl, o("wl")),
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.C), dl, o("wh")),
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (p(r.CR), dl, o("sh")),
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.CR), dh, o("sl"), p(r.CR)),
        "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.C), dh, o("wl"), p(r.C)),
    ] + ([] if h2 else [
        "v_lshlrev_b32 %s, 1, %s" % (v(r.X), dh),
    ]) + [
        "v_lshrrev_b64 %s, 31, %s" % (p(r.CR), p(r.CR)),
        ("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.QE), dh, o("sh2"), p(r.CR))) if h2 else
        ("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (p(r.QE), v(r.X), o("sh"), p(r.CR))),  # the last read of a per-lane twiddle
        "v_mad_u64_u32 %s, vcc, %s, %%[nqh], %s" % (p(r.C), v(r.QE), p(r.C)),
        "v_mad_u64_u32 %s, vcc, %s, %%[nql], %s" % (p(r.C), v(r.QE + 1), p(r.C)),
        "v_add_u32 %s, %s, %s" % (v(r.ACC + 1), v(r.ACC + 1), v(r.C)),                # dl wl + (c << 32)
        "v_mad_u64_u32 %s, vcc, %s, %%[nql], %s" % (o("bo"), v(r.QE), p(r.ACC)),      # b' = d w: the last product lands in the output
    ]


def interleave(streams):
    out = []
    for k in range(max(len(s) for s in streams)):
        for s in streams:
            if k < len(s):
                out.append(s[k])
    return out


def emit(name, gs, nb, h2=False):
    """h2: the twiddle's doubled high Shoup word is an operand (t.sh2)"""
    lines = interleave([(gs_stream if gs else ct_stream)(i, h2) for i in range(nb)])
    body = " \\\n".join('        "%s\\n\\t"' % ln for ln in lines)
    # forward: no early-clobber -- every input is read before the first output is written (checked below), so outputs may reuse
    # input registers.  inverse: a' is written before the twiddles are read -> its two halves are early-clobber outputs.
    early = ("aol", "aoh") if gs else ()
    late = ("bo",) if gs else ("ao", "bol", "boh")
    first_out = min(k for k, ln in enumerate(lines) if any("%%[%s" % nm in ln for nm in late))
    vgpr_inputs = ["%%[%s%d]" % (nm, i) for nm in ("a", "b", "al", "ah", "bl", "bh", "wl", "wh", "sl", "sh", "sh2") for i in range(nb)]
    for ln in lines[first_out + 1:]:
        assert not any(op + "," in ln + "," or ln.endswith(op) for op in vgpr_inputs), "input read after an output was written: " + ln
    # VCC: written by the low half of a difference, read by its high half two or more instructions later, untouched in between
    for k, ln in enumerate(lines):
        if ln.startswith("v_sub_co_u32"):
            j = next(j for j in range(k + 1, len(lines)) if lines[j].startswith("v_subb_co_u32"))
            assert j - k - 1 >= 2 and not any("vcc" in x for x in lines[k + 1:j]), "VCC hazard: " + ln
    if gs:
        outs = ", ".join('[aol%d] "=&v"(aol%d), [aoh%d] "=&v"(aoh%d), [bo%d] "=v"(bo%d)' % ((i,) * 6) for i in range(nb))
    else:
        outs = ", ".join('[ao%d] "=v"(ao%d), [bol%d] "=v"(bol%d), [boh%d] "=v"(boh%d)' % ((i,) * 6) for i in range(nb))
    twc = "TWC"
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
    ins.append('[nql] "s"(m.nql), [nqh] "s"(m.nqh), [nq4] "s"(m.nq4), [q4] "s"(m.q4)')
    clob = ['"vcc"'] + ['"v%d"' % x for i in range(nb) for x in Regs(i, gs).all()]
    return ("#define %s(TWC) \\\n    asm( \\\n%s \\\n        : %s \\\n        : %s \\\n        : %s)\n"
            % (name, body, outs, ", \\\n          ".join(ins), ", ".join(clob)))


def main():
    text = ("// ntt16_bfly.inc -- GENERATED by tools/gen_ntt16_bfly.py; do not edit.  See that script and ntt16_kernel.h.\n"
            "// NTT16_CT1(TWC): one forward butterfly on (a0, b0, t0) -> (ao0, {bol0, boh0}); NTT16_GS1(TWC): one inverse butterfly\n"
            "// -> ({aol0, aoh0}, bo0).  TWC = NTT16_S (wave-uniform twiddles, SGPR operands) or NTT16_V (per-lane twiddles); m = ModC.\n"
            "// NTT16_CT1H(TWC) / NTT16_GS1H(TWC): 2 sh is an operand of its own (t0.sh2) -- one instruction fewer.\n")
    # one butterfly per block: measured, a wave issues at most every other VALU slot whatever its instruction-level parallelism
    # (interleaving two butterflies per block changed nothing but the register count), so parallelism comes from waves
    text += emit("NTT16_CT1", False, 1) + emit("NTT16_GS1", True, 1) + emit("NTT16_CT1H", False, 1, True) + emit("NTT16_GS1H", True, 1, True)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
