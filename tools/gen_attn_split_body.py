"""Generator of the software-pipelined iteration body of attn_fwd_split_kernel (ladcast_amd/csrc/attn_split.hip).

Writes ladcast_amd/csrc/attn_split_body_t3.inc (split-bf16, 3 MFMAs per product) and attn_split_body_t1.inc (single-term bf16):
the statement list of ONE iteration - 48 (16) MFMAs of S_next = K_{t+1}.Q^T with the row sums and the hi/lo split of the current
tile's probabilities and the LDS-DMA issues in their gaps, the rare rescale, then 48 (16) MFMAs of O^T += V_t^T.P^T with the
running max and the exponentials of S_next in their gaps.  The counted `s_waitcnt lgkmcnt(N)` in front of every fragment use is
computed here from the order in which the fragment reads are issued (LDS reads return in order), so a placement change cannot
leave a stale count behind.  The .inc files are checked in: the build does not run this script.

usage: python tools/gen_attn_split_body.py
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Body:
    def __init__(self, terms):
        self.terms = terms
        self.out = []
        self.reads = []  # tags of the LDS read INSTRUCTIONS issued so far, in order

    def e(self, s=""):
        self.out.append("      " + s)

    def sb(self):
        self.e("LDC_SB;")

    # -- fragment reads ---------------------------------------------------------------------------------------------------
    def rd_k(self, slot, a):  # pair a = (kt, s): kt = a & 1, s = a >> 1
        kt, s = a & 1, a >> 1
        off = s * 4096 + kt * 2048
        self.e(f"LDC_RDK(kh{slot}, k_hi, {off});")
        self.reads.append(("k", slot))
        if self.terms == 3:
            self.e(f"LDC_RDK(kl{slot}, k_lo, {off});")
            self.reads.append(("k", slot))

    def rd_v(self, slot, dt):
        for hilo in range(2 if self.terms == 3 else 1):
            n = "vh" if hilo == 0 else "vl"
            for kt in range(2):
                off = ((hilo * 2 + kt) * 8 + dt) * 512
                self.e(f"LDC_RDV({n}{slot}{'ab'[kt]}, v_ad, {off});")
                self.reads.append(("v", slot))

    def wait(self, kind, slot):
        """lgkmcnt(N): N = read instructions issued after the last read of (kind, slot)"""
        idx = max(i for i, t in enumerate(self.reads) if t == (kind, slot))
        n = len(self.reads) - 1 - idx
        if kind == "k":
            regs = f'"+v"(kh{slot})' + (f', "+v"(kl{slot})' if self.terms == 3 else "")
        else:
            regs = f'"+v"(vh{slot}a), "+v"(vh{slot}b)' + (f', "+v"(vl{slot}a), "+v"(vl{slot}b)' if self.terms == 3 else "")
        self.e(f'asm volatile("s_waitcnt lgkmcnt({n})" : {regs});')
        self.sb()

    # -- MFMAs ----------------------------------------------------------------------------------------------------------------
    def mm_qk(self, kind, slot, kt, qt, s, first):
        a = {"lh": f"kl{slot}", "hl": f"kh{slot}", "hh": f"kh{slot}"}[kind]
        b = {"lh": f"qh[{qt}][{s}]", "hl": f"ql[{qt}][{s}]", "hh": f"qh[{qt}][{s}]"}[kind]
        c = f"c0[{kt}]" if first else f"sn[{kt}][{qt}]"  # c0: zero, or the additive key bias of the tile (scale_attn_by_lat)
        self.e(f"sn[{kt}][{qt}] = LDC_MFMA(LDC_BF(({a})), {b}, {c});")
        self.sb()

    def mm_pv(self, kind, slot, dt, qt):
        a = {"lh": f"LDC_CAT(vl{slot}a, vl{slot}b)", "hl": f"LDC_CAT(vh{slot}a, vh{slot}b)", "hh": f"LDC_CAT(vh{slot}a, vh{slot}b)"}[kind]
        b = {"lh": f"ph[{qt}]", "hl": f"pl[{qt}]", "hh": f"ph[{qt}]"}[kind]
        self.e(f"o[{dt}][{qt}] = LDC_MFMA(LDC_BF(({a})), LDC_BF(({b})), o[{dt}][{qt}]);")
        self.sb()


def generate(terms):
    g = Body(terms)
    e = g.e
    kinds = ("lh", "hl", "hh") if terms == 3 else ("hh",)
    # ---- VALU pieces ------------------------------------------------------------------------------------------------------
    # phase A: row sums of the current tile's probabilities ec[kt][qt][r] (per-lane partial sums: the cross-lane part is done once,
    # after the last tile), running l, and the hi / lo split into the P fragments
    piecesA = []
    for qt in range(2):
        piecesA.append(f"t0 = ec[0][{qt}][0] + ec[0][{qt}][1]; t1 = ec[0][{qt}][2] + ec[0][{qt}][3]; t2 = ec[1][{qt}][0] + ec[1][{qt}][1]; "
                       f"t3 = ec[1][{qt}][2] + ec[1][{qt}][3]; asm volatile(\"\" : \"+v\"(t0), \"+v\"(t1), \"+v\"(t2), \"+v\"(t3));")
        piecesA.append(f"t0 += t1; t2 += t3; t0 += t2; l_run[{qt}] = l_run[{qt}] * alpha[{qt}] + t0; asm volatile(\"\" : \"+v\"(l_run[{qt}]));")
    for qt in range(2):
        for i in range(4):  # pair i of the fragment: keys (kt = i >> 1, r = 2 (i & 1), +1)
            kt, r = i >> 1, 2 * (i & 1)
            piecesA.append(f"LDC_SPLIT_PAIR(ph[{qt}], pl[{qt}], {i}, ec[{kt}][{qt}][{r}], ec[{kt}][{qt}][{r + 1}])")
    dma = [f"dma_k({j});" for j in range(4)] + [f"dma_v({j});" for j in range(4)]  # unconditional: no branch in the MFMA stream (tile index clamped)
    # phase B: running max (lazy) and exponentials of S_next -> the next iteration's ec / alpha
    piecesB = []
    piecesB.append("if (mask_next) { LDC_MASK_TAIL(sn) }")
    for qt in range(2):
        piecesB.append(f"mx[{qt}] = max3f(max3f(sn[0][{qt}][0], sn[0][{qt}][1], sn[0][{qt}][2]), max3f(sn[0][{qt}][3], sn[1][{qt}][0], sn[1][{qt}][1]), "
                       f"fmaxf(sn[1][{qt}][2], sn[1][{qt}][3])); asm volatile(\"\" : \"+v\"(mx[{qt}]));")
    for qt in range(2):
        piecesB.append(f"mx[{qt}] = xor16_max(mx[{qt}]); asm volatile(\"\" : \"+v\"(mx[{qt}]));")
        piecesB.append(f"mx[{qt}] = xor32_max(mx[{qt}]); asm volatile(\"\" : \"+v\"(mx[{qt}]));")
    for qt in range(2):
        # lazy running max (scores are in log2 units): it only moves when exceeded by more than 2^8, so the rescale of O is rare;
        # the last iteration's S_next is computed from a stale stage - any bit pattern, NaN and inf included - and must not move it:
        # the compare is ANDed (bitwise: no short-circuit branch) with the wave-uniform live_next; one compare + one select per query
        piecesB.append(f"m_new[{qt}] = ((mx[{qt}] - m_run[{qt}] > 8.0f) & live_next) ? mx[{qt}] : m_run[{qt}]; "
                       f"alpha_n[{qt}] = __builtin_amdgcn_exp2f(m_run[{qt}] - m_new[{qt}]); m_run[{qt}] = m_new[{qt}]; "
                       f"asm volatile(\"\" : \"+v\"(m_new[{qt}]), \"+v\"(alpha_n[{qt}]));")
    for kt in range(2):
        for qt in range(2):
            for h in range(2):
                r = 2 * h
                piecesB.append(f"sn[{kt}][{qt}][{r}] = __builtin_amdgcn_exp2f(sn[{kt}][{qt}][{r}] - m_new[{qt}]); "
                               f"sn[{kt}][{qt}][{r + 1}] = __builtin_amdgcn_exp2f(sn[{kt}][{qt}][{r + 1}] - m_new[{qt}]); "
                               f"asm volatile(\"\" : \"+v\"(sn[{kt}][{qt}][{r}]), \"+v\"(sn[{kt}][{qt}][{r + 1}]));")

    # ---- phase A ------------------------------------------------------------------------------------------------------------
    n_mm = 8 * 2 * len(kinds)  # MFMAs per phase
    e(f"// ---- phase A: S_next = K_(t+1) . Q^T ({n_mm} MFMAs); row sums + split of the current tile's P and the DMA issues in the gaps ----")
    for a in range(4):
        g.rd_k(a, a)
    g.sb()
    # gap plan: pieces first (the split must be complete before phase B), then the DMA issues; spread evenly, several per gap if
    # there are more pieces than gaps (single-term mode)
    planA = piecesA + dma
    gaps = {}
    for i, p in enumerate(planA):
        gaps.setdefault((i * n_mm) // len(planA), []).append(p)
    gi = 0
    for a in range(8):
        slot, kt, s = a & 3, a & 1, a >> 1
        g.wait("k", slot)
        for ki, kind in enumerate(kinds):
            for qt in range(2):
                g.mm_qk(kind, slot, kt, qt, s, first=(s == 0 and ki == 0))
                last = (ki == len(kinds) - 1 and qt == 1)
                if last:
                    if a < 4:
                        g.rd_k(slot, a + 4)
                    else:
                        g.rd_v(slot, a - 4)
                for p_ in gaps.get(gi, []):
                    e(p_)
                    g.sb()
                gi += 1
    assert all(k < gi for k in gaps), (max(gaps), gi)
    e("// ---- rare: the running max moved at this tile -> rescale O ----")
    e("if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f)) {")
    e("#pragma unroll")
    e("  for (int d = 0; d < 8; ++d)")
    e("#pragma unroll")
    e("    for (int r = 0; r < 4; ++r) { o[d][0][r] *= alpha[0]; o[d][1][r] *= alpha[1]; }")
    e("}")
    g.sb()
    # ---- phase B ------------------------------------------------------------------------------------------------------------
    e(f"// ---- phase B: O^T += V_t^T . P^T ({n_mm} MFMAs); running max + exponentials of S_next in the gaps ----")
    gapsB = {}
    for i, p in enumerate(piecesB):  # S_next's last MFMA is a few instructions back: start one gap in
        gapsB.setdefault(1 + (i * (n_mm - 2)) // len(piecesB), []).append(p)
    assert max(gapsB) < n_mm
    gi = 0
    for dt in range(8):
        slot = dt & 3
        g.wait("v", slot)
        for ki, kind in enumerate(kinds):
            for qt in range(2):
                g.mm_pv(kind, slot, dt, qt)
                last = (ki == len(kinds) - 1 and qt == 1)
                if last and dt < 4:
                    g.rd_v(slot, dt + 4)
                for p_ in gapsB.get(gi, []):
                    e(p_)
                    g.sb()
                gi += 1
    return "\n".join(g.out) + "\n"


if __name__ == "__main__":
    for terms in (3, 1):
        path = os.path.join(ROOT, "ladcast_amd", "csrc", f"attn_split_body_t{terms}.inc")
        hdr = f"// GENERATED by tools/gen_attn_split_body.py (TERMS = {terms}); do not edit - regenerate.\n"
        with open(path, "w") as f:
            f.write(hdr + generate(terms))
        print("wrote", path)
