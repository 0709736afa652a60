#!/usr/bin/env python3
"""Generator of the hand-scheduled Poseidon-Goldilocks permutation for gfx950 (csrc/poseidon_asm.inc).

The compiler's output for the permutation spends a third of its issue slots on register-pair moves, carry chains
(v_sub_co / v_subb_co cost 2.6 simple-op slots each on this chip) and compare/select fix-ups.  This
script emits the permutation as ONE inline-asm statement with its own register allocation and interleaving:

  * lanes are pairs of 32-bit digits (x = x0 + x1 phi, phi = 2^32, phi^2 = phi - 1 mod p), any representative < 2^64;
  * a product is 4 v_mad_u64_u32 + 3 moves (exact code: + a select that re-injects the carry of the third product through the
    high half of the last addend); its reduction is w0 + w1 phi + w2 (phi - 1) - w3 with ONE multiply-add, one 64-bit
    subtraction, three scalar mask operations, three selects and one 64-bit add - exact for every input; the fast code
    (mul_task) is 13 instructions: the carry of the third product is the borrow-in of the final 64-bit subtraction;
  * the MDS layer is 288 v_mad_u64_u32 into 24 64-bit digit sums whose initial values are the next round's constants (no
    constant is ever "added"), an output al + ah phi is folded exactly with one multiply-add (al + ah1 (phi - 1)), one add
    with carry-out, one select and one 64-bit add;
  * the S-boxes of a full round run three at a time, the S-box of a partial round is interleaved with the 176
    multiply-adds of the layer that do not depend on it;
  * every value inside the statement is "any representative below 2^64"; nothing is approximate and there is no fall-back.

Two statements are generated from the same round code: POSEIDON_ASM_PERMUTE (one permutation of twelve 64-bit operands, round
0's constants already added) and POSEIDON_ASM_SPONGE (the whole hash_no_pad loop of the Merkle leaf kernel: loads of the
next eight columns, round 0's constants, permutation; the state never leaves the register block).

The instruction lists are first executed by the interpreter below (one lane, Python integers) on the known-answer vectors
and random inputs and compared with the textbook permutation; then printed.
usage: python tools/gen_poseidon_asm.py [out_dir = plonky2_bn254_amd/csrc]
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from derive_poseidon_constants import KATS, MDS_CIRC, P, permute, round_constants  # noqa: E402

M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1

# ---- register map (physical registers named inside the asm string; all of them are declared clobbered) ----------------
VB = 26            # first VGPR of the block; the block ends at v126 (128 registers = four waves per SIMD)
A0 = VB            # bank A: the state, lane i = v[A0+2i : A0+2i+1]
S0 = VB + 24       # bank S: S-box outputs of a full round / the second state bank of the partial rounds
TMP = VB + 48      # S-box stream temporaries / MDS digit sums
N_STREAMS = 3
STREAM_REGS = 16
E0 = TMP + N_STREAMS * STREAM_REGS     # two pairs (e, 0): the "+ (2^32 - 1) if carry" addends of the exact code
MN_A = TMP + N_STREAMS * STREAM_REGS - 2    # fast code: running min of the low digits al0 (fold needs al0 >= 2^10 > ah1) ...
MX_A = TMP + N_STREAMS * STREAM_REGS - 1    # ... running max of ah0 (fold needs ah0 + al1 + ah1 < 2^32): the two registers the last
MN_R = E0 + 4                          # stream does not use (its X3 aliases X2); (unused since the products' check became scalar: S_FLAG)
V_END = E0 + 5
MODE = {"fast": True}
A = [(A0 + 2 * i, A0 + 2 * i + 1) for i in range(12)]
S = [(S0 + 2 * i, S0 + 2 * i + 1) for i in range(12)]
LD0 = 10           # v10..v25: staging of the next chunk's eight loads (sponge statement only)
SB_LOOP = 24       # s[24:29]: column pointer, column stride in bytes, element index, leaf length (sponge statement)
S_BLK = 20         # s[20:21]: pointer to the records of the merged partial-round blocks
SB_MASK = (30, 34, 22)    # s[30:31], s[34:35], s[22:23]: one scratch mask pair per stream (s32, s33 stay untouched)
S_FLAG = 30               # fast code: s[30:31] (a mask pair only the exact code uses) = sticky underflow flag of the products
SB_CARRY = 36      # s[36:47]: two carry pairs per stream; the folds use the first four pairs
SB_CONST = 48      # s[48:95]: the 48 dwords of one round of initial digit sums (or round 0's 12 constants)
S_PTR = 96         # s[96:97]: running pointer into the table
S_CNT = 98
S_HALF = 99
ROUND_BYTES = 192


def init_table(rc):
    """INIT[r][j]: what the MDS layer of round r adds to output j = the constants of round r's own partial-round lanes pushed
    through the layer + the pre-S-box constant of round r + 1 (all 12 lanes before a full round, lane 0 before a partial
    one).  Round 0's own pre-S-box constants are added up front.  One extra all-zero round pads the prefetch."""
    full = lambda r: r < 4 or r >= 26
    tab = []
    for r in range(30):
        row = []
        for j in range(12):
            k = 0
            if not full(r):
                k += sum(MDS_CIRC[(i - j) % 12] * rc[12 * r + i] for i in range(1, 12))
            if r + 1 < 30:
                if full(r + 1) or j == 0:
                    k += rc[12 * (r + 1) + j]
            row.append(k % P)
        tab.append(row)
    tab.append([0] * 12)
    return tab


def table_dwords(tab):
    out = []
    for row in tab:
        for k in row:
            out += [k & M32, 0, k >> 32, 0]
    return out


# ---- instruction emission ---------------------------------------------------------------------------------------------
class Prog:
    def __init__(self):
        self.ins = []      # tuples (op, args...)

    def emit(self, *t):
        self.ins.append(t)


def v(i):
    return ("v", i)


def vp(i):
    assert i % 2 == 0, i
    return ("vp", i)


def sp(i):
    assert i % 2 == 0
    return ("sp", i)


def s1(i):
    return ("s", i)


def interleave(p, tasks):
    for t in round_robin(tasks):
        p.emit(*t)


def round_robin(tasks):
    tasks = list(tasks)
    while tasks:
        for t in list(tasks):
            try:
                yield next(t)
            except StopIteration:
                tasks.remove(t)


def mul_task(a, b, dst, t, cA, cB, cT):
    """(a0,a1) x (b0,b1) -> dst pair, any representative below 2^64 (exact code: always; fast code: unless the running min
    S_FLAG is raised).  a, b: (lo reg, hi reg); t: base of 10 stream
    temporaries with v[t+3] == 0 throughout; cA, cB, cT: three SGPR pairs of the stream.  Yields instructions one at a time so
    that several products can be interleaved.
    product (w0..w3 32-bit words) = P0 | R0 | S0 | S1 from four multiply-adds; reduction X = w0 + w1 phi + w2 (phi - 1) - w3:
    t = V + w2 (2^32 - 1) with carry c, t' = t - w3 with borrow b (two carry instructions), then + (c - b)(2^32 - 1):
    after a carry t < 2^64 - 2^33 and after a borrow t' > 2^64 - 2^32, so neither correction wraps again."""
    a0, a1 = a
    b0, b1 = b
    Pp, T, Q, R, Sx = t, t + 2, t + 4, t + 6, t + 8
    if MODE["fast"]:
        # 13 instructions.  v[t+5] == 0 throughout as well: the carry of the third product is not put into the high half of the
        # last addend (a select) but taken as the borrow-in of the final subtraction, whose two borrow instructions replace
        # the select / subtract / sign-extension / 64-bit add of the 14-instruction form:
        #   t = V + w2 (2^32 - 1) (carry c), t += c ? 2^32 - 1 : 0 (cannot wrap), dst = t - w3 - carry3 as a 64-bit subtraction.
        # Not covered: c = 0 and t < w3 + carry3 (an underflow): that is exactly the borrow-out of the last instruction, which a
        # scalar OR collects in S_FLAG for the check after the permutation (no vector instruction spent on the check).
        yield ("mad", vp(Pp), "vcc", v(a0), v(b0), 0)
        yield ("mov", v(T), v(Pp + 1))
        yield ("mad", vp(Sx), "vcc", v(a0), v(b1), vp(T))
        yield ("mad", vp(R), sp(cA), v(a1), v(b0), vp(Sx))     # carry3 -> cA, kept until the subtraction
        yield ("mov", v(Q), v(R + 1))                      # U = (R1, 0)
        yield ("mov", v(Pp + 1), v(R))                     # V = (w0, w1)
        yield ("mad", vp(Sx), "vcc", v(a1), v(b1), vp(Q))  # (w2, w3 without carry3)
        yield ("mad", vp(R), sp(cB), v(Sx), -1, vp(Pp))    # t = V + w2 * (2^32 - 1), carry c -> cB
        yield ("cnd", v(Q), 0, -1, sp(cB))                 # e
        yield ("add64", vp(R), vp(R), vp(Q))               # t + e
        yield ("subbco", v(dst), sp(cA), v(R), v(Sx + 1), sp(cA))
        yield ("subbco", v(dst + 1), sp(cA), v(R + 1), 0, sp(cA))
        yield ("s_or", sp(S_FLAG), sp(S_FLAG), sp(cA))     # the final borrow IS the underflow: sticky flag (a scalar instruction)
        return
    yield ("mad", vp(Pp), "vcc", v(a0), v(b0), 0)
    yield ("mov", v(T), v(Pp + 1))
    yield ("mad", vp(Q), "vcc", v(a0), v(b1), vp(T))
    yield ("mad", vp(R), sp(cA), v(a1), v(b0), vp(Q))
    yield ("mov", v(Q), v(R + 1))                      # U.lo = R1
    yield ("mov", v(Pp + 1), v(R))                     # V = (w0, w1)
    yield ("cnd", v(Q + 1), 0, 1, sp(cA))              # U.hi = carry of the third product
    yield ("mad", vp(Sx), "vcc", v(a1), v(b1), vp(Q))  # S = (w2, w3)
    yield ("mad", vp(R), sp(cB), v(Sx), -1, vp(Pp))    # t = V + w2 * (2^32 - 1), carry c -> cB
    yield ("subco", v(R), sp(cA), v(R), v(Sx + 1))     # t' = t - w3 ...
    yield ("subbco", v(R + 1), sp(cA), v(R + 1), 0, sp(cA))   # ... borrow b -> cA
    yield ("s_xor", sp(cT), sp(cA), sp(cB))
    yield ("s_and", sp(cB), sp(cT), sp(cB))            # c and not b: + (2^32 - 1)
    yield ("s_and", sp(cA), sp(cT), sp(cA))            # b and not c: - (2^32 - 1) = + (1, 0xFFFFFFFF)
    yield ("cnd", v(Q), 0, -1, sp(cB))
    yield ("cnd", v(Q + 1), 0, -1, sp(cA))
    yield ("cnd", v(Q), v(Q), 1, sp(cA))
    yield ("add64", vp(dst), vp(R), vp(Q))


def sbox_task(x, dst, stream):
    """x^7 of the lane held in the registers x = (lo, hi) -> the pair dst (dst may be x's own pair)."""
    t = TMP + stream * STREAM_REGS
    X2, X3, X4 = t + 10, t + 10, t + 12      # x^3 = x^2 * x is written (last instruction) where x^2 was read (first ones)
    cA, cB, cT = SB_CARRY + 4 * stream, SB_CARRY + 4 * stream + 2, SB_MASK[stream]
    yield from mul_task(x, x, X2, t, cA, cB, cT)
    yield from mul_task((X2, X2 + 1), (X2, X2 + 1), X4, t, cA, cB, cT)
    yield from mul_task((X2, X2 + 1), x, X3, t, cA, cB, cT)
    yield from mul_task((X3, X3 + 1), (X4, X4 + 1), dst, t, cA, cB, cT)


def mds_terms(j, lane0_last):
    """(coefficient, source lane) of output j, optionally with the lane-0 terms at the end."""
    terms = [(MDS_CIRC[i] + (8 if i == j == 0 else 0), (i + j) % 12) for i in range(12)]     # (the diagonal 8 s_0 merged into row 0)
    if lane0_last:
        terms = [t for t in terms if t[1] != 0] + [t for t in terms if t[1] == 0]
    return terms


def mds_group_tasks(src, g, lane0_last, split_at_lane0=False, acc0=TMP):
    """Accumulation of outputs 4g..4g+3: eight independent chains (al, ah of four outputs) as eight tasks.  With
    split_at_lane0 every chain is returned as (before, after) the first lane-0 term."""
    tasks = []
    for q in range(4):
        j = 4 * g + q
        for h in range(2):
            acc = acc0 + 4 * q + 2 * h
            init = SB_CONST + 4 * j + 2 * h

            def chain(j=j, h=h, acc=acc, init=init, part=None):
                first = True
                for c, lane in mds_terms(j, lane0_last):
                    is0 = lane == 0
                    if part == "before" and is0:
                        return
                    if part == "after" and not is0:
                        first = False
                        continue
                    yield ("mad", vp(acc), "vcc", v(src[lane][h]), c, sp(init) if first else vp(acc))
                    first = False
            if split_at_lane0:
                tasks.append((chain(part="before"), chain(part="after")))
            else:
                tasks.append(chain())
    return tasks


def fold_group(p, dst, g, acc0=TMP):
    """out_j = al + ah phi exactly, any representative: u = al + ah1 (phi - 1) (no carry: both below 2^42);
    (u1 + ah0) mod 2^32 with its carry c; + c (2^32 - 1) (cannot wrap: after a carry the high digit is below 2^10)."""
    q_al = [acc0 + 4 * q for q in range(4)]
    q_ah = [acc0 + 4 * q + 2 for q in range(4)]
    if MODE["fast"]:
        # (al0 - ah1) + (al1 + ah0 + ah1) phi: two instructions per output plus one running min / max per two outputs; not
        # covered: a borrow of the low digit or a carry of the high one, excluded when al0 >= 2^10 and ah0 < 2^32 - 2^10
        # (al1, ah1 <= 284)
        for q in range(4):
            d = dst[4 * g + q]
            p.emit("sub", v(d[0]), v(q_al[q]), v(q_ah[q] + 1))
            p.emit("add3", v(d[1]), v(q_al[q] + 1), v(q_ah[q]), v(q_ah[q] + 1))
        for q in (0, 2):
            p.emit("min3", v(MN_A), v(MN_A), v(q_al[q]), v(q_al[q + 1]))
            p.emit("max3", v(MX_A), v(MX_A), v(q_ah[q]), v(q_ah[q + 1]))
        return
    for q in range(4):
        p.emit("mad", vp(q_al[q]), "vcc", v(q_ah[q] + 1), -1, vp(q_al[q]))
    for q in range(4):
        p.emit("addco", v(q_al[q] + 1), sp(SB_CARRY + 2 * q), v(q_al[q] + 1), v(q_ah[q]))
    for half in (0, 2):
        for k in (0, 1):
            p.emit("cnd", v(E0 + 2 * k), 0, -1, sp(SB_CARRY + 2 * (half + k)))
        for k in (0, 1):
            p.emit("add64", vp(dst[4 * g + half + k][0]), vp(q_al[half + k]), vp(E0 + 2 * k))


def rezero_stream_temps(p, streams):
    """The MDS digit sums overlay the S-box temporaries: restore the zero high halves of the T pairs."""
    for s in streams:
        p.emit("mov", v(TMP + s * STREAM_REGS + 3), 0)
        if MODE["fast"]:
            p.emit("mov", v(TMP + s * STREAM_REGS + 5), 0)     # the high half of the U pair (mul_task, fast code)


def prefetch_next(p):
    p.emit("s_add_ptr", ROUND_BYTES)
    for k in range(3):
        p.emit("s_load16", SB_CONST + 16 * k, 64 * k)


EXPERIMENT = os.environ.get("POSEIDON_GEN_EXPERIMENT", "")     # timing experiments only (tools/ubench): wrong results


def full_round(p):
    """A -> S-boxes -> S -> MDS -> A."""
    rezero_stream_temps(p, range(N_STREAMS))
    if "nosbox" not in EXPERIMENT:
        for k in range(0, 12, N_STREAMS):
            interleave(p, [sbox_task(A[lane], S[lane][0], s) for s, lane in enumerate(range(k, min(12, k + N_STREAMS)))])
    p.emit("s_waitcnt")
    if "nomds" in EXPERIMENT:
        prefetch_next(p)
        return
    for g in range(3):
        interleave(p, mds_group_tasks(S, g, False))
        if g == 2:
            prefetch_next(p)          # every initial digit sum of this round has been consumed
        if "nofold" not in EXPERIMENT:
            fold_group(p, A, g)


def partial_round(p, src, dst, ratio=3):
    """The single S-box (lane 0, in the last stream's registers, result in place) is interleaved with the part of output
    groups 0 and 1 that does not depend on lane 0 (176 multiply-adds), `ratio` of them per S-box instruction; the lane-0
    terms, the folds and group 2 follow."""
    assert N_STREAMS * STREAM_REGS >= 32 + STREAM_REGS
    p.emit("s_waitcnt")
    head = sbox_task(src[0], src[0][0], N_STREAMS - 1)
    acc1 = TMP + 16
    pairs = mds_group_tasks(src, 0, True, split_at_lane0=True) + mds_group_tasks(src, 1, True, split_at_lane0=True, acc0=acc1)
    body = round_robin([b for b, _ in pairs])
    head_done = body_done = False
    while not (head_done and body_done):
        if not head_done:
            try:
                p.emit(*next(head))
            except StopIteration:
                head_done = True
        for _ in range(ratio):
            if not body_done:
                try:
                    p.emit(*next(body))
                except StopIteration:
                    body_done = True
    interleave(p, [a for _, a in pairs])
    fold_group(p, dst, 0)
    fold_group(p, dst, 1, acc0=acc1)
    interleave(p, mds_group_tasks(src, 2, False))
    prefetch_next(p)
    fold_group(p, dst, 2)


# ---- three partial rounds per linear layer ------------------------------------------------------------------------------------
# In a partial round only lane 0 is non-linear: with M' = M with column 0 removed, m0 = column 0 of M and t = S-box output,
#   y = M' u + t m0 + k.  Three rounds composed:
#   y0   = row0(M') u + t0 m0[0] + k0[0]                           -> t1 = sbox(y0)
#   y'0  = row0(M'^2) u + t0 (M' m0)[0] + t1 m0[0] + (M' k0 + k1)[0]  -> t2 = sbox(y'0)
#   y''  = M'^3 u + t0 M'^2 m0 + t1 M' m0 + t2 m0 + (M'^2 k0 + M' k1 + k2)
# The integer matrices M'^2, M'^3 have entries below 2^15 / 2^24, so the 32-bit-digit sums still fit 64-bit accumulators:
# 264 multiply-adds replace three layers of 290, the three S-boxes stay sequential but run beside the 308 multiply-adds that
# do not depend on them.  Constants per block come from POSEIDON_BLK_DEV (15 records of 16 dwords, block_tables()).
BLK_BYTES = 15 * 64


def mat_M():
    return [[MDS_CIRC[(i - j) % 12] + (8 if i == j == 0 else 0) for i in range(12)] for j in range(12)]


def mat_mul(X, Y):
    return [[sum(X[i][k] * Y[k][j] for k in range(12)) for j in range(12)] for i in range(12)]


def mat_vec(X, v_, mod=None):
    r = [sum(X[i][k] * v_[k] for k in range(12)) for i in range(12)]
    return [x % mod for x in r] if mod else r


def block_matrices():
    M = mat_M()
    Mp = [[0 if i == 0 else M[j][i] for i in range(12)] for j in range(12)]
    m0 = [M[j][0] for j in range(12)]
    Mp2 = mat_mul(Mp, Mp)
    Mp3 = mat_mul(Mp2, Mp)
    return M, Mp, Mp2, Mp3, m0, mat_vec(Mp, m0), mat_vec(Mp2, m0)


def block_tables(tab):
    """Records of the 7 blocks (rounds 4+3b .. 6+3b).  tab = init_table(rc): tab[r] is what round r's layer adds."""
    M, Mp, Mp2, Mp3, m0, b, a = block_matrices()
    assert max(sum(r) for r in Mp3) + max(a) + max(b) + 64 < 1 << 31        # every digit sum stays below 2^63
    out = []
    for blk in range(7):
        r = 4 + 3 * blk
        k0, k1, k2 = tab[r], tab[r + 1], tab[r + 2]
        ky0 = k0[0]
        ky1 = (mat_vec(Mp, k0, P)[0] + k1[0]) % P
        kk = [(x + y + z) % P for x, y, z in zip(mat_vec(Mp2, k0, P), mat_vec(Mp, k1, P), k2)]
        pad = lambda c: [c & M32, 0, c >> 32, 0]
        out += [Mp2[0][i] for i in range(1, 12)] + [b[0]] + pad(ky0)          # record 0
        out += pad(ky1) + a                                                    # record 1
        out += b + [0, 0, 0, 0]                                                # record 2
        for j in range(12):
            out += [Mp3[j][i] for i in range(1, 12)] + [0] + pad(kk[j])       # output records
    return out


def fold_exact(al, ah, carry, e, dst):
    """dst = al + ah phi, exact (see fold_group); al, ah, dst: register pairs (dst may be al); e: an (x, 0) pair."""
    yield ("mad", vp(al), "vcc", v(ah + 1), -1, vp(al))
    yield ("addco", v(al + 1), sp(carry), v(al + 1), v(ah))
    yield ("cnd", v(e), 0, -1, sp(carry))
    yield ("add64", vp(dst), vp(al), vp(e))


def partial_block(p, src, dst, ratio=2):
    M, Mp, Mp2, Mp3, m0, b, a = block_matrices()
    C0, C1, C2 = SB_CONST, SB_CONST + 16, SB_CONST + 32          # three 16-dword scalar slots
    AH = [TMP + 2 * j for j in range(12)]                        # high-digit sums of the 12 outputs; low-digit sums in dst
    AL = [dst[j][0] for j in range(12)]
    Y0L, Y0H, Y1L, Y1H = TMP + 24, TMP + 26, TMP + 28, TMP + 30
    stream = N_STREAMS - 1
    cfold = SB_CARRY                                             # carry pairs of the folds (streams 0, 1 are idle here)

    def chain():   # the sequential part: three S-boxes and the two lane-0 values between them
        yield from sbox_task(src[0], src[0][0], stream)                      # t0 in place of u0
        yield ("await", "y0")
        yield ("mad", vp(Y0L), "vcc", v(src[0][0]), m0[0], vp(Y0L))
        yield ("mad", vp(Y0H), "vcc", v(src[0][1]), m0[0], vp(Y0H))
        yield from fold_exact(Y0L, Y0H, cfold, E0, Y0L)
        yield from sbox_task((Y0L, Y0L + 1), Y0L, stream)                    # t1 in place of y0
        yield ("await", "y1")
        yield ("mad", vp(Y1L), "vcc", v(src[0][0]), s1(C0 + 11), vp(Y1L))
        yield ("mad", vp(Y1H), "vcc", v(src[0][1]), s1(C0 + 11), vp(Y1H))
        yield ("mad", vp(Y1L), "vcc", v(Y0L), m0[0], vp(Y1L))
        yield ("mad", vp(Y1H), "vcc", v(Y0L + 1), m0[0], vp(Y1H))
        yield ("release", "c0")                                              # slot C0 may be overwritten from here on
        yield from fold_exact(Y1L, Y1H, cfold + 2, E0 + 2, Y1L)
        yield from sbox_task((Y1L, Y1L + 1), Y1L, stream)                    # t2 in place of y'0

    def bulk():    # everything that does not depend on the S-boxes
        for h, acc in ((0, Y0L), (1, Y0H)):
            for i in range(1, 12):
                yield ("mad", vp(acc), "vcc", v(src[i][h]), M[0][i], sp(C0 + 12 + 2 * h) if i == 1 else vp(acc))
        yield ("mark", "y0")
        # (one scalar operand per vector instruction: a scalar coefficient and a scalar initial value cannot share a multiply-add)
        for h, acc in ((0, Y1L), (1, Y1H)):
            yield ("mov64", vp(acc), sp(C1 + 2 * h))
            for i in range(1, 12):
                yield ("mad", vp(acc), "vcc", v(src[i][h]), s1(C0 + i - 1), vp(acc))
        yield ("mark", "y1")
        for j in range(12):
            slot = C2 if j % 2 == 0 else C0
            if j == 1:
                yield ("need", "c0")                     # slot C0 still holds the lane-0 coefficient of y'0
            if j >= 1:
                yield ("s_load16_blk", slot, 64 * (3 + j))
            yield ("s_waitcnt",)
            if j == 0:
                pass
            for h, acc in ((0, AL[j]), (1, AH[j])):
                yield ("mov64", vp(acc), sp(slot + 12 + 2 * h))
            for i in range(1, 12):
                for h, acc in ((0, AL[j]), (1, AH[j])):
                    yield ("mad", vp(acc), "vcc", v(src[i][h]), s1(slot + i - 1), vp(acc))

    # prologue: records 0, 1 and the first output record
    p.emit("s_load16_blk", C0, 0)
    p.emit("s_load16_blk", C1, 64)
    p.emit("s_load16_blk", C2, 64 * 3)
    p.emit("s_waitcnt")
    ch, bk = chain(), bulk()
    marks, released = set(), set()
    ch_next = bk_next = None
    ch_done = bk_done = False

    def pull(g):
        try:
            return next(g)
        except StopIteration:
            return None

    ch_next, bk_next = pull(ch), pull(bk)
    while ch_next is not None or bk_next is not None:
        # one instruction of the chain (if it is not waiting for the bulk)
        if ch_next is not None:
            if ch_next[0] == "await":
                if ch_next[1] in marks:
                    ch_next = pull(ch)
                    continue
            elif ch_next[0] == "release":
                released.add(ch_next[1])
                ch_next = pull(ch)
                continue
            else:
                p.emit(*ch_next)
                ch_next = pull(ch)
        # `ratio` instructions of the bulk (if it is not waiting for the chain)
        for _ in range(ratio):
            if bk_next is None:
                break
            if bk_next[0] == "mark":
                marks.add(bk_next[1])
                bk_next = pull(bk)
                continue
            if bk_next[0] == "need":
                if bk_next[1] in released:
                    bk_next = pull(bk)
                    continue
                break
            p.emit(*bk_next)
            bk_next = pull(bk)
    # the S-box outputs enter the twelve lanes; folds
    p.emit("s_load16_blk", C0, 128)            # record 2: b
    p.emit("s_waitcnt")
    tk = [(src[0][0], src[0][1]), (Y0L, Y0L + 1), (Y1L, Y1L + 1)]

    def tail(j):
        coef = [s1(C1 + 4 + j), s1(C0 + j), m0[j]]
        for k in range(3):
            yield ("mad", vp(AL[j]), "vcc", v(tk[k][0]), coef[k], vp(AL[j]))
            yield ("mad", vp(AH[j]), "vcc", v(tk[k][1]), coef[k], vp(AH[j]))
        yield from fold_exact(AL[j], AH[j], cfold + 2 * (j % 4), E0 + 2 * (j % 2), AL[j])
    for j0 in range(0, 12, 2):        # two at a time: there are two (e, 0) pairs
        interleave(p, [tail(j) for j in range(j0, j0 + 2)])
    p.emit("s_add_blk", BLK_BYTES)


def permutation_body(p, tag=""):
    """30 rounds on bank A (round 0's constants already added); the table pointer S_PTR must point at INIT[0]."""
    p.emit("s_mov", S_HALF, 0)
    for k in range(3):
        p.emit("s_load16", SB_CONST + 16 * k, 64 * k)
    p.emit("label", "half" + tag)
    p.emit("s_mov", S_CNT, 4)
    p.emit("label", "full" + tag)
    full_round(p)
    p.emit("loop", S_CNT, "full" + tag)
    p.emit("s_branch_if_ne0", S_HALF, "done" + tag)
    rezero_stream_temps(p, [N_STREAMS - 1])
    if MODE["fast"] and "noblocks" not in EXPERIMENT:
        # rounds 4..24 as seven blocks of three, round 25 alone
        p.emit("s_waitcnt")                # the prefetch of the last full round must land before the slots are reused
        p.emit("e_zero")
        p.emit("s_mov", S_CNT, 3)
        p.emit("label", "part" + tag)
        partial_block(p, A, S)
        partial_block(p, S, A)
        p.emit("loop", S_CNT, "part" + tag)
        partial_block(p, A, S)
        p.emit("s_add_ptr", ROUND_BYTES * 21)
        for k in range(3):
            p.emit("s_load16", SB_CONST + 16 * k, 64 * k)
        partial_round(p, S, A)
    else:
        p.emit("s_mov", S_CNT, 11)
        p.emit("label", "part" + tag)
        partial_round(p, A, S)
        partial_round(p, S, A)
        p.emit("loop", S_CNT, "part" + tag)
    p.emit("s_mov", S_HALF, 1)
    p.emit("s_branch", "half" + tag)
    p.emit("label", "done" + tag)
    p.emit("s_waitcnt")                    # the last prefetch (padding round) must land before the registers are reused


def flags_init(p):
    p.emit("mov", v(MN_A), -1)
    p.emit("mov", v(MX_A), 0)
    p.emit("s_mov", S_FLAG, 0)
    p.emit("s_mov", S_FLAG + 1, 0)


def build_permute():
    """Fast code first; if its running min / max show that some lane left the range the short forms cover (about one
    wave-permutation in 100), the operands are copied in again and the exact code runs."""
    p = Prog()
    p.emit("copy_in")                      # v_mov_b64 of the 12 operands into bank A
    flags_init(p)
    p.emit("s_ptr", "tab")
    p.emit("s_ptr_blk")
    MODE["fast"] = True
    permutation_body(p)
    p.emit("flagcheck", "ok")
    p.emit("copy_in")
    p.emit("e_zero")
    p.emit("s_ptr", "tab")
    MODE["fast"] = False
    permutation_body(p, tag="x")
    MODE["fast"] = True
    p.emit("label", "ok")
    p.emit("copy_out")
    return p


def build_sponge(store=False):
    """hash_no_pad over leaf_len elements (leaf_len > 0) from the all-zero state: per chunk of <= 8 elements overwrite lanes
    0.., add round 0's constants (lazily: any representative), permute.  The loads of chunk c + 1 are issued before the
    permutation of chunk c into staging registers; the input of every permutation is parked in LDS for the (rare) exact
    repeat."""
    p = Prog()
    p.emit("zero_in")
    p.emit("sponge_init")                  # s[24:25] = column pointer, s[26:27] = stride, s28 = 0, s29 = leaf_len
    p.emit("s_rem")
    for i in range(8):
        p.emit("gload", vp(LD0 + 2 * i), i, 0)     # if (i < len) LD[i] = *(col + lane offset); col += stride
    p.emit("label", "chunk")
    p.emit("s_rem")                        # S_CNT = len - idx
    p.emit("s_waitcnt_all")
    for i in range(8):
        p.emit("take", vp(A[i][0]), vp(LD0 + 2 * i), i)   # if (idx + i < len) A[i] = LD[i]
    for i in range(8):
        p.emit("gload", vp(LD0 + 2 * i), i, 8)     # if (idx + 8 + i < len) LD[i] = next element
    p.emit("s_ptr", "rc")
    for k in range(3):
        p.emit("s_load8", SB_CONST + 8 * k, 32 * k)
    p.emit("e_zero")
    p.emit("s_waitcnt")
    # lazy addition of round 0's constants: s = a + rc; wrapped iff s < rc; then + (2^32 - 1)
    for base in range(0, 12, 2):
        for k in (0, 1):
            i = base + k
            p.emit("add64", vp(A[i][0]), vp(A[i][0]), sp(SB_CONST + 2 * i))
        for k in (0, 1):
            i = base + k
            p.emit("cmplt64", sp(SB_CARRY + 2 * k), vp(A[i][0]), sp(SB_CONST + 2 * i))
        p.emit("s_nop", 0)
        for k in (0, 1):
            p.emit("cnd", v(E0 + 2 * k), 0, -1, sp(SB_CARRY + 2 * k))
        for k in (0, 1):
            i = base + k
            p.emit("add64", vp(A[i][0]), vp(A[i][0]), vp(E0 + 2 * k))
    p.emit("lds_save")
    flags_init(p)
    p.emit("s_ptr", "tab")
    p.emit("s_ptr_blk")
    MODE["fast"] = True
    permutation_body(p)
    p.emit("flagcheck", "next")
    p.emit("lds_restore")
    p.emit("e_zero")
    p.emit("s_ptr", "tab")
    MODE["fast"] = False
    permutation_body(p, tag="x")
    MODE["fast"] = True
    p.emit("label", "next")
    p.emit("chunk_loop", "chunk")          # idx += 8; if (idx < len) goto chunk
    p.emit("digest_store" if store else "digest_out")
    return p


# ---- hazards: a VALU instruction may read an SGPR written by a VALU instruction only 2 wait states later --------------------
VALU = ("mad", "mov", "mov64", "cnd", "sub", "add3", "min", "min3", "max3", "add64", "addco", "subco", "subbco", "cmplt64")


def sgpr_reads(t):
    srcs = t[3:] if t[0] in ("mad", "addco", "subco", "subbco") else t[2:]
    return [a[1] for a in srcs if isinstance(a, tuple) and a[0] in ("sp", "s")]


def sgpr_write(t):
    if t[0] in ("mad", "addco", "subco", "subbco") and isinstance(t[2], tuple):
        return t[2][1]
    if t[0] == "cmplt64":
        return t[1][1]
    return None


def pad_hazards(ins):
    out = []
    last_write = {}
    n = 0
    for t in ins:
        if t[0] in ("label", "loop", "s_branch", "s_branch_if_ne0", "chunk_loop", "flagcheck"):
            last_write.clear()       # nothing is assumed across control flow: every target starts with instructions that
            #                          read no VALU-written SGPR within two slots (checked by assert_targets_safe)
        if t[0] in VALU:
            need = 0
            for r in sgpr_reads(t):
                if r in last_write:
                    need = max(need, 2 - (n - last_write[r] - 1))
            if need > 0:
                out.append(("s_nop", need - 1))
                n += need
            w = sgpr_write(t)
            if w is not None:
                last_write[w] = n
        out.append(t)
        n += 1
    return out


def assert_targets_safe(ins):
    """After a label the first two VALU instructions must not read a carry / mask SGPR (their writer may be the last
    instruction before the branch)."""
    for k, t in enumerate(ins):
        if t[0] == "label":
            seen = 0
            for u in ins[k + 1:]:
                if u[0] in VALU:
                    assert not [r for r in sgpr_reads(u) if r < SB_CONST], (t, u)
                    seen += 1
                    if seen == 2:
                        break


# ---- one-lane interpreter -------------------------------------------------------------------------------------------------
class Machine:
    def __init__(self):
        self.v = {}
        self.s = {}
        self.pending = set()      # SGPRs written by a scalar load that no s_waitcnt has covered yet

    def rd(self, a):
        if isinstance(a, int):
            return a & M64 if a >= 0 else a & M32     # inline constants: -1 is 0xFFFFFFFF as a 32-bit source
        k, i = a
        if k == "v":
            return self.v[i]
        if k == "vp":
            return self.v[i] | (self.v[i + 1] << 32)
        if k == "sp":
            assert i not in self.pending and i + 1 not in self.pending, "scalar load not waited for: s%d" % i
            return self.s[i] | (self.s[i + 1] << 32)
        if k == "s":
            assert i not in self.pending, "scalar load not waited for: s%d" % i
            return self.s[i]
        raise ValueError(a)

    def wr(self, a, val):
        k, i = a
        if k == "v":
            self.v[i] = val & M32
        elif k == "vp":
            self.v[i], self.v[i + 1] = val & M32, (val >> 32) & M32
        elif k == "sp":
            self.s[i], self.s[i + 1] = val & M32, (val >> 32) & M32


def run(ins, mem, state, leaf=None, stats=None):
    """mem: {"tab": dwords, "rc": dwords}; state: 12 lanes copied in; leaf: the lane's elements (sponge statement);
    stats: counts the flag checks and exact repeats ("force": always repeat)."""
    stats = {} if stats is None else stats
    saved = None
    m = Machine()
    labels = {t[1]: k for k, t in enumerate(ins) if t[0] == "label"}
    pc = steps = 0
    blk_ptr = 0
    ptr = ("tab", 0)
    col = 0
    while pc < len(ins):
        t = ins[pc]
        op = t[0]
        pc += 1
        steps += 1
        if "ops" in stats:
            stats["ops"][op] = stats["ops"].get(op, 0) + 1
        if op == "copy_in":
            for i, x in enumerate(state):
                m.wr(vp(A[i][0]), x)
        elif op == "e_zero":
            m.v[E0 + 1] = m.v[E0 + 3] = 0
        elif op == "flagcheck":
            stats["checks"] = stats.get("checks", 0) + 1
            if not (m.v[MN_A] < 1024 or m.v[MX_A] > 0xFFFFFBFF or (m.s[S_FLAG] | m.s[S_FLAG + 1]) != 0 or stats.get("force")):
                pc = labels[t[1]]
            else:
                stats["repeats"] = stats.get("repeats", 0) + 1
        elif op == "lds_save":
            saved = [m.rd(vp(A[i][0])) for i in range(12)]
        elif op == "lds_restore":
            for i in range(12):
                m.wr(vp(A[i][0]), saved[i])
        elif op == "copy_out":
            break
        elif op == "zero_in":
            for i in range(12):
                m.wr(vp(A[i][0]), 0)
        elif op in ("digest_out", "digest_store"):
            break
        elif op == "sponge_init":
            m.s[SB_LOOP + 4], m.s[SB_LOOP + 5] = 0, len(leaf)
        elif op == "s_rem":
            m.s[S_CNT] = m.s[SB_LOOP + 5] - m.s[SB_LOOP + 4]
        elif op == "gload":
            if m.s[S_CNT] > t[2] + t[3]:
                m.wr(t[1], leaf[col])
                col += 1
        elif op == "take":
            if m.s[S_CNT] > t[3]:
                m.wr(t[1], m.rd(t[2]))
        elif op == "chunk_loop":
            m.s[SB_LOOP + 4] += 8
            if m.s[SB_LOOP + 4] < m.s[SB_LOOP + 5]:
                pc = labels[t[1]]
        elif op == "mad":
            _, d, c, x, y, z = t
            a, b = m.rd(x), m.rd(y)
            assert a <= M32 and b <= M32
            r = a * b + (m.rd(z) if z != 0 else 0)
            m.wr(d, r & M64)
            assert (r >> 64) <= 1
            if c != "vcc":
                m.wr(c, r >> 64)      # the lane's bit of the carry mask
            else:
                assert r >> 64 == 0, "carry into the scratch destination would be lost"
        elif op == "addco":
            r = m.rd(t[3]) + m.rd(t[4])
            m.wr(t[1], r & M32)
            m.wr(t[2], r >> 32)
        elif op == "subco":
            a, b = m.rd(t[3]), m.rd(t[4])
            m.wr(t[1], (a - b) & M32)
            m.wr(t[2], 1 if a < b else 0)
        elif op == "subbco":
            a, b, c = m.rd(t[3]), m.rd(t[4]), m.rd(t[5]) & 1
            m.wr(t[1], (a - b - c) & M32)
            m.wr(t[2], 1 if a < b + c else 0)
        elif op == "s_xor":
            m.wr(t[1], m.rd(t[2]) ^ m.rd(t[3]))
        elif op == "s_and":
            m.wr(t[1], m.rd(t[2]) & m.rd(t[3]))
        elif op == "s_or":
            m.wr(t[1], m.rd(t[2]) | m.rd(t[3]))
        elif op == "cmplt64":
            m.wr(t[1], 1 if m.rd(t[2]) < m.rd(t[3]) else 0)
        elif op in ("mov", "mov64"):
            m.wr(t[1], m.rd(t[2]))
        elif op == "cnd":
            m.wr(t[1], m.rd(t[3]) if m.rd(t[4]) & 1 else m.rd(t[2]))
        elif op == "sub":
            m.wr(t[1], (m.rd(t[2]) - m.rd(t[3])) & M32)
        elif op == "add3":
            m.wr(t[1], (m.rd(t[2]) + m.rd(t[3]) + m.rd(t[4])) & M32)
        elif op == "min":
            m.wr(t[1], min(m.rd(t[2]), m.rd(t[3])))
        elif op == "min3":
            m.wr(t[1], min(m.rd(t[2]), m.rd(t[3]), m.rd(t[4])))
        elif op == "max3":
            m.wr(t[1], max(m.rd(t[2]), m.rd(t[3]), m.rd(t[4])))
        elif op == "add64":
            m.wr(t[1], (m.rd(t[2]) + m.rd(t[3])) & M64)
        elif op == "s_mov":
            m.s[t[1]] = t[2]
        elif op == "s_ptr":
            ptr = (t[1], 0)
        elif op == "s_add_ptr":
            ptr = (ptr[0], ptr[1] + t[1])
        elif op in ("s_load16", "s_load8"):
            n = 16 if op == "s_load16" else 8
            base = (ptr[1] + t[2]) // 4
            for k in range(n):
                m.s[t[1] + k] = mem[ptr[0]][base + k]
                m.pending.add(t[1] + k)
        elif op == "s_load16_blk":
            base = (blk_ptr + t[2]) // 4
            for k in range(16):
                m.s[t[1] + k] = mem["blk"][base + k]
                m.pending.add(t[1] + k)
        elif op == "s_add_blk":
            blk_ptr += t[1]
        elif op == "s_ptr_blk":
            blk_ptr = 0
        elif op in ("s_waitcnt", "s_waitcnt_all"):
            m.pending.clear()
        elif op in ("label", "s_nop"):
            pass
        elif op == "loop":
            m.s[t[1]] -= 1
            if m.s[t[1]] != 0:
                pc = labels[t[2]]
        elif op == "s_branch":
            pc = labels[t[1]]
        elif op == "s_branch_if_ne0":
            if m.s[t[1]] != 0:
                pc = labels[t[2]]
        else:
            raise ValueError(op)
    out = [m.rd(vp(A[i][0])) for i in range(12)]
    return out, False, steps


# ---- printing ---------------------------------------------------------------------------------------------------------------
def fmt(a):
    if isinstance(a, int):
        return str(a)
    if a == "vcc":
        return "vcc"
    k, i = a
    return {"v": "v%d" % i, "vp": "v[%d:%d]" % (i, i + 1), "sp": "s[%d:%d]" % (i, i + 1), "s": "s%d" % i}[k]


def text(ins):
    L = []
    for t in ins:
        op = t[0]
        if op == "copy_in":
            for i in range(12):
                L.append("v_mov_b64 v[%d:%d], %%[x%d]" % (A[i][0], A[i][1], i))
        elif op == "e_zero":
            L.append("v_mov_b32 v%d, 0" % (E0 + 1))
            L.append("v_mov_b32 v%d, 0" % (E0 + 3))
        elif op == "flagcheck":
            L.append("v_cmp_gt_u32 vcc, 0x400, v%d" % MN_A)
            L.append("s_mov_b64 s[%d:%d], vcc" % (SB_CARRY, SB_CARRY + 1))
            L.append("v_cmp_lt_u32 vcc, 0xfffffbff, v%d" % MX_A)
            L.append("s_or_b64 s[%d:%d], s[%d:%d], vcc" % (SB_CARRY, SB_CARRY + 1, SB_CARRY, SB_CARRY + 1))
            L.append("s_or_b64 vcc, s[%d:%d], s[%d:%d]" % (S_FLAG, S_FLAG + 1, SB_CARRY, SB_CARRY + 1))
            L.append("s_cbranch_vccz Lpos_%s_%%=" % t[1])
        elif op == "lds_save":
            for i in range(12):
                L.append("ds_write_b64 %%[lds], v[%d:%d] offset:%d" % (A[i][0], A[i][1], 2048 * i))
        elif op == "lds_restore":
            for i in range(12):
                L.append("ds_read_b64 v[%d:%d], %%[lds] offset:%d" % (A[i][0], A[i][1], 2048 * i))
            L.append("s_waitcnt lgkmcnt(0)")
        elif op == "copy_out":
            for i in range(12):
                L.append("v_mov_b64 %%[x%d], v[%d:%d]" % (i, A[i][0], A[i][1]))
        elif op == "sponge_init":
            L.append("s_mov_b64 s[%d:%d], %%[col]" % (SB_LOOP, SB_LOOP + 1))
            L.append("s_mov_b64 s[%d:%d], %%[stride]" % (SB_LOOP + 2, SB_LOOP + 3))
            L.append("s_mov_b32 s%d, 0" % (SB_LOOP + 4))
            L.append("s_mov_b32 s%d, %%[len]" % (SB_LOOP + 5))
        elif op == "zero_in":
            for i in range(12):
                L.append("v_mov_b64 v[%d:%d], 0" % (A[i][0], A[i][1]))
        elif op == "digest_out":
            for i in range(4):
                L.append("v_mov_b64 %%[o%d], v[%d:%d]" % (i, A[i][0], A[i][1]))
        elif op == "digest_store":
            # canonical digest (a >= p: a - p = a + 2^32 - 1 mod 2^64) -> 32 bytes at dig + doff; the staging registers are free now
            assert all(A[i][1] == A[i][0] + 1 and A[i + 1][0] == A[i][0] + 2 for i in range(3))
            L.append("s_mov_b64 s[%d:%d], 0xffffffff" % (SB_CARRY, SB_CARRY + 1))
            L.append("s_mov_b32 s%d, 0" % (SB_CARRY + 2))
            L.append("s_mov_b32 s%d, -1" % (SB_CARRY + 3))          # s[+2:+3] = p - 1
            for i in range(4):
                tl = LD0 + 2 * i
                L.append("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (tl, tl + 1, A[i][0], A[i][1], SB_CARRY, SB_CARRY + 1))
            for i in range(4):
                tl = LD0 + 2 * i
                L.append("v_cmp_lt_u64 vcc, s[%d:%d], v[%d:%d]" % (SB_CARRY + 2, SB_CARRY + 3, A[i][0], A[i][1]))
                L.append("s_nop 1")
                L.append("v_cndmask_b32 v%d, v%d, v%d, vcc" % (A[i][0], A[i][0], tl))
                L.append("v_cndmask_b32 v%d, v%d, v%d, vcc" % (A[i][1], A[i][1], tl + 1))
            L.append("global_store_dwordx4 %%[doff], v[%d:%d], %%[dig]" % (A[0][0], A[1][1]))
            L.append("global_store_dwordx4 %%[doff], v[%d:%d], %%[dig] offset:16" % (A[2][0], A[3][1]))
        elif op == "s_rem":       # remaining = len - idx; element idx + k exists iff remaining > k
            L.append("s_sub_u32 s%d, s%d, s%d" % (S_CNT, SB_LOOP + 5, SB_LOOP + 4))
        elif op == "gload":
            k = t[2] + t[3]
            skip = "Lpos_skip%d_%d_%d_%%=" % (len(L), t[2], t[3])
            L.append("s_cmp_gt_u32 s%d, %d" % (S_CNT, k))
            L.append("s_cbranch_scc0 " + skip)
            L.append("global_load_dwordx2 %s, %%[off], s[%d:%d]" % (fmt(t[1]), SB_LOOP, SB_LOOP + 1))
            L.append("s_add_u32 s%d, s%d, s%d" % (SB_LOOP, SB_LOOP, SB_LOOP + 2))
            L.append("s_addc_u32 s%d, s%d, s%d" % (SB_LOOP + 1, SB_LOOP + 1, SB_LOOP + 3))
            L.append(skip + ":")
        elif op == "take":
            skip = "Lpos_keep%d_%d_%%=" % (len(L), t[3])
            L.append("s_cmp_gt_u32 s%d, %d" % (S_CNT, t[3]))
            L.append("s_cbranch_scc0 " + skip)
            L.append("v_mov_b64 %s, %s" % (fmt(t[1]), fmt(t[2])))
            L.append(skip + ":")
        elif op == "chunk_loop":
            L.append("s_add_u32 s%d, s%d, 8" % (SB_LOOP + 4, SB_LOOP + 4))
            L.append("s_cmp_lt_u32 s%d, s%d" % (SB_LOOP + 4, SB_LOOP + 5))
            L.append("s_cbranch_scc1 Lpos_%s_%%=" % t[1])
        elif op == "mad":
            L.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "addco":
            L.append("v_add_co_u32 %s, %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "cmplt64":
            L.append("v_cmp_lt_u64 %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "mov":
            L.append("v_mov_b32 %s, %s" % (fmt(t[1]), fmt(t[2])))
        elif op == "mov64":
            L.append("v_mov_b64 %s, %s" % (fmt(t[1]), fmt(t[2])))
        elif op == "cnd":
            L.append("v_cndmask_b32 %s, %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "sub":
            L.append("v_sub_u32 %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "add3":
            L.append("v_add3_u32 %s, %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "min":
            L.append("v_min_u32 %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op in ("min3", "max3"):
            L.append("v_%s_u32 %s, %s, %s, %s" % tuple([op] + [fmt(a) for a in t[1:]]))
        elif op == "add64":
            L.append("v_lshl_add_u64 %s, %s, 0, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "subco":
            L.append("v_sub_co_u32 %s, %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "subbco":
            L.append("v_subb_co_u32 %s, %s, %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "s_xor":
            L.append("s_xor_b64 %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "s_and":
            L.append("s_and_b64 %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "s_or":
            L.append("s_or_b64 %s, %s, %s" % tuple(fmt(a) for a in t[1:]))
        elif op == "s_mov":
            L.append("s_mov_b32 s%d, %d" % (t[1], t[2]))
        elif op == "s_ptr":
            L.append("s_mov_b64 s[%d:%d], %%[%s]" % (S_PTR, S_PTR + 1, t[1]))
        elif op == "s_add_ptr":
            L.append("s_add_u32 s%d, s%d, %d" % (S_PTR, S_PTR, t[1]))
            L.append("s_addc_u32 s%d, s%d, 0" % (S_PTR + 1, S_PTR + 1))
        elif op == "s_load16":
            L.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x%x" % (t[1], t[1] + 15, S_PTR, S_PTR + 1, t[2]))
        elif op == "s_load16_blk":
            L.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x%x" % (t[1], t[1] + 15, S_BLK, S_BLK + 1, t[2]))
        elif op == "s_add_blk":
            L.append("s_add_u32 s%d, s%d, %d" % (S_BLK, S_BLK, t[1]))
            L.append("s_addc_u32 s%d, s%d, 0" % (S_BLK + 1, S_BLK + 1))
        elif op == "s_ptr_blk":
            L.append("s_mov_b64 s[%d:%d], %%[blk]" % (S_BLK, S_BLK + 1))
        elif op == "s_load8":
            L.append("s_load_dwordx8 s[%d:%d], s[%d:%d], 0x%x" % (t[1], t[1] + 7, S_PTR, S_PTR + 1, t[2]))
        elif op == "s_waitcnt":
            L.append("s_waitcnt lgkmcnt(0)")
        elif op == "s_waitcnt_all":
            L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        elif op == "s_nop":
            L.append("s_nop %d" % t[1])
        elif op == "label":
            L.append("Lpos_%s_%%=:" % t[1])
        elif op == "loop":
            L.append("s_sub_u32 s%d, s%d, 1" % (t[1], t[1]))
            L.append("s_cmp_lg_u32 s%d, 0" % t[1])
            L.append("s_cbranch_scc1 Lpos_%s_%%=" % t[2])
        elif op == "s_branch":
            L.append("s_branch Lpos_%s_%%=" % t[1])
        elif op == "s_branch_if_ne0":
            L.append("s_cmp_lg_u32 s%d, 0" % t[1])
            L.append("s_cbranch_scc1 Lpos_%s_%%=" % t[2])
        else:
            raise ValueError(op)
    return L


# measured issue cost in cycles per wave instruction and SIMD (tools/ubench/sgpr_ops.hip): two classes; the full-rate ones
# overlap with the half-rate ones of the same wave
HALF = {"mov64": 4.15, "mad": 4.15, "cnd": 4.15, "add64": 4.15, "addco": 4.15, "subco": 4.15, "subbco": 4.15, "cmplt64": 4.15, "add3": 4.15,
        "min3": 4.15, "max3": 4.15}
FULL = {"mov": 2.1, "sub": 2.1, "min": 2.1}
UNIT = dict(HALF, **FULL)


def dynamic_counts(ins):
    labels = {t[1]: k for k, t in enumerate(ins) if t[0] == "label"}
    loops = [k for k, t in enumerate(ins) if t[0] == "loop"]

    def count(lo, hi):
        c = {}
        for t in ins[lo:hi]:
            c[t[0]] = c.get(t[0], 0) + 1
        return c
    full_c = count(labels["full"], loops[0])
    part_c = count(labels["part"], loops[1])
    dyn = {k: 8 * full_c.get(k, 0) + 11 * part_c.get(k, 0) for k in set(full_c) | set(part_c)}
    valu = sum(n for k, n in dyn.items() if k in UNIT)
    half = sum(n for k, n in dyn.items() if k in HALF)
    return valu, half, 4.15 * half + 3.6 * (valu - half)     # measured: in this mix the full-rate ones do not overlap


def write_macro(f, name, lines):
    f.write("#define %s \\\n" % name)
    for ln in lines:
        f.write('  "%s\\n" \\\n' % ln)
    f.write('  ""\n')


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                                 "plonky2_bn254_amd", "csrc")
    rc = round_constants()
    tab = init_table(rc)
    mem = {"tab": table_dwords(tab), "rc": [w for c in rc[:12] for w in (c & M32, c >> 32)], "blk": block_tables(tab)}
    if EXPERIMENT:
        ins = pad_hazards(build_permute().ins)
        with open(os.path.join(out_dir, "poseidon_asm_%s.inc" % EXPERIMENT.replace(",", "_")), "w") as f:
            write_macro(f, "POSEIDON_ASM_EXP_" + EXPERIMENT.replace(",", "_").upper(), text(ins))
        print("experiment", EXPERIMENT, dynamic_counts(ins))
        return
    perm = pad_hazards(build_permute().ins)
    sponge = pad_hazards(build_sponge().ins)
    sponge_store = pad_hazards(build_sponge(store=True).ins)
    assert_targets_safe(perm)
    assert_targets_safe(sponge)
    assert_targets_safe(sponge_store)

    # ---- check: interpreter vs the textbook permutation ----
    def check(state):
        pre = [(x + rc[i]) % P for i, x in enumerate(state)]     # the caller adds round 0's constants
        got, bad, steps = run(perm, mem, pre, stats=stats)
        assert [g % P for g in got] == permute(state, rc), state
        got, _, _ = run(perm, mem, pre, stats={"force": 1})          # the exact code alone
        assert [g % P for g in got] == permute(state, rc), state
        return bad, steps
    rnd = random.Random(1)
    stats = {}
    n_bad = steps = 0
    for inp, _ in KATS:
        assert not check(inp)[0]
    for _ in range(int(os.environ.get("POSEIDON_GEN_TESTS", "60"))):
        st = [rnd.randrange(P) for _ in range(12)]
        if rnd.random() < 0.3:
            st = [rnd.choice([0, 1, P - 1, (1 << 32) - 1, 1 << 32, rnd.randrange(1 << 16)]) for _ in range(12)]
        b, steps = check(st)
        n_bad += b
    # sponge: hash_no_pad of ragged leaves from non-canonical-looking state words
    for n in (1, 5, 8, 9, 16, 21):
        leaf = [rnd.randrange(P) if rnd.random() < 0.7 else rnd.choice([0, P - 1, 1]) for _ in range(n)]
        got, bad, _ = run(sponge, mem, None, leaf, stats={"force": n % 2})
        got2, _, _ = run(sponge_store, mem, None, leaf, stats={"force": n % 2})
        assert got2 == got
        st = [0] * 12
        for c in range(0, n, 8):
            st[:len(leaf[c:c + 8])] = leaf[c:c + 8]
            st = permute(st, rc)
        assert [g % P for g in got[:4]] == st[:4], n
    st1 = {"ops": {}}
    run(perm, mem, [(x + rc[i]) % P for i, x in enumerate(KATS[1][0])], stats=st1)
    valu = sum(n for k, n in st1["ops"].items() if k in UNIT)
    half = sum(n for k, n in st1["ops"].items() if k in HALF)
    cycles = 4.15 * half + 3.6 * (valu - half)
    print("interpreter ok (%d instructions executed per permutation; exact repeats in the tests: %d of %d checks)" %
          (steps, stats.get("repeats", 0), stats.get("checks", 0)))
    print("fast code: %d VALU instructions per permutation, %d of them half-rate -> about %.0f cycles per wave and SIMD; "
          "VGPRs v%d..v%d" % (valu, half, cycles, VB, V_END - 1))

    with open(os.path.join(out_dir, "poseidon_asm.inc"), "w") as f:
        f.write("// Generated by tools/gen_poseidon_asm.py - do not edit.  Hand-scheduled Poseidon-Goldilocks for gfx950; see the\n"
                "// generator for the algorithm.  %d VALU instructions per permutation.\n"
                "// POSEIDON_ASM_PERMUTE: x0..x11 = the twelve lanes (64-bit, in/out, round 0's constants already added, any\n"
                "//   representative below 2^64 comes back), tab = POSEIDON_INIT_DEV.\n"
                "// POSEIDON_ASM_SPONGE: hash_no_pad of one leaf per lane from the zero state: o0..o3 = digest (out, any\n"
                "//   representative), col = address of element 0 of lane offset 0, off = the lane's byte offset (32 bits), stride =\n"
                "//   bytes between consecutive elements, len = leaf length (> 0), rc = POSEIDON_RC_DEV, tab as above; also\n"
                "//   clobbers POSEIDON_ASM_SPONGE_CLOBBERS (the staging registers of the next chunk's loads).\n" % valu)
        f.write("#define POSEIDON_ASM_VGPR_FIRST %d\n#define POSEIDON_ASM_VGPR_LAST %d\n" % (VB, V_END - 1))
        write_macro(f, "POSEIDON_ASM_PERMUTE", text(perm))
        write_macro(f, "POSEIDON_ASM_SPONGE", text(sponge))
        f.write("// POSEIDON_ASM_SPONGE_STORE: the same, but the CANONICAL digest is written to memory by the statement itself (32 bytes at\n"
                "//   dig + doff: dig = scalar base address, doff = the lane's byte offset, 32 bits) instead of coming back in o0..o3: the\n"
                "//   statement leaves the compiler ten VGPRs, and four 64-bit outputs beside its inputs made it spill to scratch memory.\n")
        write_macro(f, "POSEIDON_ASM_SPONGE_STORE", text(sponge_store))
        f.write("#define POSEIDON_ASM_SPONGE_CLOBBERS " + ", ".join('"v%d"' % i for i in range(LD0, LD0 + 16)) + "\n")
        clob = ['"s%d"' % S_BLK, '"s%d"' % (S_BLK + 1)] + ['"v%d"' % i for i in range(VB, V_END)] + ['"s%d"' % i for i in list(range(SB_LOOP, SB_LOOP + 6)) + list(range(SB_CARRY, S_HALF + 1)) + [30, 31, 34, 35, 22, 23]] + ['"vcc"', '"scc"']
        f.write("#define POSEIDON_ASM_CLOBBERS " + ", ".join(clob) + "\n")
    with open(os.path.join(out_dir, "poseidon_init.inc"), "w") as f:
        f.write("/* Initial digit sums of the MDS layer of every round (tools/gen_poseidon_asm.py: init_table): per round and output\n"
                "   lane four dwords (low 32 bits, 0, high 32 bits, 0) of the constant the layer adds; 31 rounds (the last is padding). */\n")
        d = mem["tab"]
        for i in range(0, len(d), 8):
            f.write("  " + ", ".join("0x%08xu" % x for x in d[i:i + 8]) + ",\n")
    with open(os.path.join(out_dir, "poseidon_blocks.inc"), "w") as f:
        f.write("/* Records of the merged partial-round blocks (tools/gen_poseidon_asm.py: block_tables): 7 blocks x 15 records x 16\n"
                "   dwords - integer coefficients of row 0 of M'^2, of M'^3 and of the S-box output vectors, and the constants. */\n")
        d = mem["blk"]
        for i in range(0, len(d), 8):
            f.write("  " + ", ".join("0x%08xu" % x for x in d[i:i + 8]) + ",\n")
    print("wrote poseidon_asm.inc (%d + %d lines), poseidon_init.inc (%d dwords)" % (len(text(perm)), len(text(sponge)), len(mem["tab"])))


if __name__ == "__main__":
    main()
