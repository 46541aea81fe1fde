"""Generator of the software-pipelined iteration body of attn_fwd_packed_kernel (ladcast_amd/csrc/attn_dma_bf16x3.hip):
prints the statement list (MFMAs with the softmax / split VALU pieces and the LDS-DMA issues hand-placed in their gaps) that was
pasted into the kernel.  Kept so the placement can be regenerated after a change; not part of the build."""
# generates the pipelined iteration body of attn_fwd_packed_kernel (pasted into attn_dma_bf16x3.hip)
slots = ["0", "1", "2", "3"]
out = []
def e(s=""): out.append("      " + s)
SB = "LDC_SB;"
def mfma_qk(kind, st, slot, first):
    a = {"lh": f"fl{slot}", "hl": f"fh{slot}", "hh": f"fh{slot}"}[kind]
    b = {"lh": f"qh[{st}]", "hl": f"ql[{st}]", "hh": f"qh[{st}]"}[kind]
    c = "zero16" if first else "sx"
    return f"sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, {a}), {b}, {c}, 0, 0, 0); {SB}"
def mfma_pv(kind, dd, slot, tt):
    a = {"lh": f"fl{slot}", "hl": f"fh{slot}", "hh": f"fh{slot}"}[kind]
    b = {"lh": f"ph{tt}", "hl": f"pl{tt}", "hh": f"ph{tt}"}[kind]
    return f"o[{dd}] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, {a}), __builtin_bit_cast(bf16x8, {b}), o[{dd}], 0, 0, 0); {SB}"
# VALU pieces
def exp_piece(j):
    return (f"sc[{2*j}] = __builtin_amdgcn_exp2f(sc[{2*j}] - m_new); sc[{2*j+1}] = __builtin_amdgcn_exp2f(sc[{2*j+1}] - m_new); "
            f"asm volatile(\"\" : \"+v\"(sc[{2*j}]), \"+v\"(sc[{2*j+1}]));")
sum_pieces = [
    "r0 = sc[0] + sc[1]; r1 = sc[2] + sc[3]; r2 = sc[4] + sc[5]; r3 = sc[6] + sc[7]; asm volatile(\"\" : \"+v\"(r0), \"+v\"(r1), \"+v\"(r2), \"+v\"(r3));",
    "r4 = sc[8] + sc[9]; r5 = sc[10] + sc[11]; r6 = sc[12] + sc[13]; r7 = sc[14] + sc[15]; asm volatile(\"\" : \"+v\"(r4), \"+v\"(r5), \"+v\"(r6), \"+v\"(r7));",
    "r0 += r1; r2 += r3; r4 += r5; r6 += r7; asm volatile(\"\" : \"+v\"(r0), \"+v\"(r2), \"+v\"(r4), \"+v\"(r6));",
    "r0 += r2; r4 += r6; r0 += r4; r0 = xor32_add(r0); l_run = l_run * alpha + r0; m_run = m_new; asm volatile(\"\" : \"+v\"(l_run));",
]
def split_piece(tt, i):  # pair i (0..3) of half tt
    a, b = f"sc[{8*tt+2*i}]", f"sc[{8*tt+2*i+1}]"
    return f"LDC_SPLIT_PAIR(ph{tt}, pl{tt}, {i}, {a}, {b})"
gapsA = {}
for j in range(8): gapsA[j] = exp_piece(j)
for j in range(4): gapsA[8 + j] = sum_pieces[j]
for i in range(4): gapsA[12 + i] = split_piece(0, i)
for j in range(4): gapsA[16 + j] = f'if (kq) dma_k({j});'
for j in range(3): gapsA[20 + j] = f'if (vq) dma_v({j});'
gapsA[23] = 'if (vq) { dma_v(3); if (wave < 2) dma_v(4); } if (kq && wave == 0) dma_k(4);'
gapsB = {0: split_piece(1, 0), 2: split_piece(1, 1), 4: split_piece(1, 2), 6: split_piece(1, 3)}
# ---- phase A ----
e("// ---- phase A: S_next = K_{t+1} . Q^T (24 MFMAs) with the softmax of S_cur in the MFMA gaps ----")
for s_ in range(4): e(f"LDC_RD_K(fh{s_}, fl{s_}, {s_})")
e("LDC_SB;")
e("{")
e("  float m0 = max3f(sc[0], sc[1], sc[2]), m1 = max3f(sc[3], sc[4], sc[5]), m2 = max3f(sc[6], sc[7], sc[8]), m3 = max3f(sc[9], sc[10], sc[11]);")
e("  const float m4 = max3f(sc[12], sc[13], sc[14]);")
e("  m0 = max3f(m0, m1, m2); m3 = max3f(m3, m4, sc[15]); m0 = fmaxf(m0, m3);")
e("  m0 = xor32_max(m0);")
e("  m_new = fmaxf(m_run, m0);")
e("  alpha = __builtin_amdgcn_exp2f(m_run - m_new);")
e("  asm volatile(\"\" : \"+v\"(m_new), \"+v\"(alpha));")
e("}")
e("LDC_SB;")
g = 0
for st in range(8):
    slot = st & 3
    e(f"LDC_W6(fh{slot}, fl{slot}); LDC_SB;")
    for kind in ("lh", "hl", "hh"):
        e(mfma_qk(kind, st, slot, st == 0 and kind == "lh"))
        if kind == "hh":
            if st < 4: e(f"LDC_RD_K(fh{slot}, fl{slot}, {st + 4})")
            else: e(f"LDC_RD_V(fh{slot}, fl{slot}, {st - 4})")
        if g in gapsA: e(gapsA[g])
        e("LDC_SB;")
        g += 1
e("// ---- rare: the running max moved -> rescale O ----")
e("if (!__all(alpha == 1.0f)) {")
e("#pragma unroll")
e("  for (int d = 0; d < 4; ++d)")
e("#pragma unroll")
e("    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;")
e("}")
e("LDC_SB;")
e("// ---- phase B: O^T += V_t^T . P^T (24 MFMAs); the second half of P is split in the gaps of the first ----")
g = 0
for J in range(8):
    slot = J & 3; tt = J >> 2; dd = J & 3
    w = {4: "LDC_W6", 5: "LDC_WN(4,", 6: "LDC_WN(2,", 7: "LDC_WN(0,"}
    if J <= 4: e(f"LDC_W6(fh{slot}, fl{slot}); LDC_SB;")
    else: e(f"{w[J]} fh{slot}, fl{slot}); LDC_SB;")
    for kind in ("lh", "hl", "hh"):
        e(mfma_pv(kind, dd, slot, tt))
        if kind == "hh" and J < 4: e(f"LDC_RD_V(fh{slot}, fl{slot}, {J + 4})")
        if g in gapsB: e(gapsB[g])
        e("LDC_SB;")
        g += 1
print("\n".join(out))
