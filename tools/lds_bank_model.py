"""LDS bank model behind the padded tile layouts of sep_bwd_kernel (backward.hip, sepb_cfg): 64 banks of 4 bytes; a ds_read_b64 is served in two
passes of 32 lanes, a ds_read_b32 in one of 64, a ds_read_b128 in four of 16; a pass takes as many cycles as the most-wanted bank has distinct
dwords.  Lane (i, q) = (lane & 15, lane >> 4) reads three 8-byte pairs at position(pixel i * S + kx) + 6 q + 2 j.  Prints the cycles of the
nine pair-reads of a 3x3 tap window for pixel strides / pads and the best layouts (ideal: 18).  python tools/lds_bank_model.py"""
def passes(addrs_dw, width):
    group = {1: 64, 2: 32, 4: 16}[width]
    tot = 0
    for g0 in range(0, 64, group):
        banks = {}
        for l in range(g0, g0 + group):
            for k in range(width):
                a = addrs_dw[l] + k
                banks.setdefault(a % 64, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot
lanes = [(l & 15, l >> 4) for l in range(64)]
def cost(stride, pad, G, S):
    tot = 0
    for kx in range(3):
        for j in range(3):
            ad = [(i * S + kx) * stride + ((i * S + kx) // G) * pad + 6 * q + 2 * j for i, q in lanes]
            tot += passes(ad, 2)
    return tot
for S in (1, 2):
    print(f"stride {S}: 24 dwords per pixel, no pad: {cost(24, 0, 8, S)} cycles for 9 pair-reads (ideal 18)")
    res = []
    for stride in (24, 28, 32):
        for G in (2, 4, 8, 16):
            for pad in (0, 4, 8, 12, 16):
                PW = 15 * S + 3
                res.append((cost(stride, pad, G, S), PW * stride + ((PW - 1) // G) * pad, stride, G, pad))
    res.sort()
    for c, row, stride, G, pad in res[:4]:
        print(f"   {c} cycles: {stride} dwords per pixel + {pad} per {G} pixels (row of {row} dwords)")
