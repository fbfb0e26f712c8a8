"""Development aid (CPU only): the LDS addressing of conv_halo_kernel (csrc/nsr_gemm_f16.hip, HaloGeo) restated in Python, to
count the bank conflicts of its A-fragment reads per geometry and tap.

Model: 64 banks x 4 B; a ds_read_b128 is served in four passes of 16 consecutive lanes; inside a pass two lanes conflict when
their 16-byte accesses touch a common bank at different addresses.  Reported: the worst and mean number of cycles a pass needs
(1 = conflict-free) over all waves / row blocks / taps of one tile.

Result (round 4, after the GPU budget was spent): the plain geometries are conflict-free; the GROUPED ones (8 images x 2
pixels per pass, image pitch = 1 mod 4) pay 2 cycles on EVERY A-fragment pass -- the counters' 1.6 conflict cycles per LDS
instruction on those kernels -- and an image pitch = 2 mod 4 with the same swizzle makes them conflict-free too (HaloGeo::pad1).

usage: python scripts/halo_lds_model.py
"""
import itertools


def geo(BN, BM, GROUPED, S):
    NI = 8 if GROUPED else 1
    TW = 8 if GROUPED else 16
    TH = 128 * BM // NI // TW
    RW = TW + 2 if S == 1 else 2 * TW + 1
    RH0 = TH + 2 if S == 1 else TH
    RH1 = TH + 2 if S == 1 else TH + 1
    pad1 = (lambda x: x + (6 - x % 4) % 4) if GROUPED else (lambda x: x)   # round 5: pitch = 2 mod 4 (round 4: 1 mod 4)
    return dict(NI=NI, TW=TW, TH=TH, RW=RW, IMG0=pad1(RW * RH0), IMG1=pad1(RW * RH1), S=S, BM=BM, GROUPED=GROUPED)


def wtap(S, j):
    return j if S == 1 else (3 + j if j < 3 else (j - 3 if j < 6 else j))


def eoff(g, j):
    t = wtap(g["S"], j)
    ky, kx = divmod(t, 3)
    if g["S"] == 1:
        return ky * g["RW"] + kx
    return (g["RW"] if ky == 2 else 0) + (g["TW"] + 1 if kx == 1 else (1 if kx == 2 else 0))


def passes_cycles(addrs):
    """cycles one 16-lane pass needs: max over banks of the number of distinct 16-byte addresses touching it"""
    per_bank = {}
    for a in addrs:
        for b in range(4):
            per_bank.setdefault(((a // 4) + b) % 64, set()).add(a)
    return max(len(v) for v in per_bank.values())


def a_fragment_conflicts(BN, BM, GROUPED, S, pitch_res=None):
    """pitch_res: grouped geometries only -- pad the image pitch to this residue mod 4 instead of the kernel's 1"""
    g = geo(BN, BM, GROUPED, S)
    if pitch_res is not None and GROUPED:
        RH0 = g["TH"] + 2 if S == 1 else g["TH"]
        RH1 = g["TH"] + 2 if S == 1 else g["TH"] + 1
        pad = lambda x: x + (pitch_res - x % 4) % 4
        g["IMG0"], g["IMG1"] = pad(g["RW"] * RH0), pad(g["RW"] * RH1)
    worst, total, n = 1, 0, 0
    for wave, bi, j, plane in itertools.product(range(4), range(BM), range(9), range(2)):
        reg = 0 if S == 1 else (0 if j < 3 else 1)
        img = g["IMG1"] if reg else g["IMG0"]
        addrs = []
        for lane in range(64):
            li, h = lane & 31, lane >> 5
            t = 32 * BM * wave + 32 * bi + li
            if GROUPED:
                dy, dx, r = t >> 6, (t >> 3) & 7, t & 7
            else:
                dy, dx, r = t >> 4, t & 15, 0
            e = r * img + dy * g["RW"] + dx + eoff(g, j)
            ad = e * 64 + ((h ^ ((e >> 2) & 3)) << 4)
            addrs.append(ad ^ (32 if plane else 0))
        for p in range(4):
            c = passes_cycles(addrs[16 * p:16 * p + 16])
            worst, total, n = max(worst, c), total + c, n + 1
    return worst, total / n


if __name__ == "__main__":
    print("A-fragment reads (ds_read_b128), cycles per 16-lane pass: worst, mean")
    for BN, BM, GROUPED, S in [(8, 2, False, 1), (8, 2, True, 1), (4, 4, False, 1), (4, 4, True, 1), (4, 2, False, 1), (4, 2, True, 1),
                               (8, 2, False, 2), (8, 2, True, 2), (4, 2, False, 2), (4, 2, True, 2)]:
        w, m = a_fragment_conflicts(BN, BM, GROUPED, S)
        print(f"  conv_halo_kernel<{BN}, {BM}, {str(GROUPED).lower()}, {S}>: worst {w}, mean {m:.2f}")
    print("grouped geometries with the image pitch = 2 mod 4 (same swizzle): worst, mean")
    for BN, BM, S in [(8, 2, 1), (4, 4, 1), (4, 2, 1), (8, 2, 2), (4, 2, 2)]:
        w, m = a_fragment_conflicts(BN, BM, True, S, pitch_res=2)
        print(f"  conv_halo_kernel<{BN}, {BM}, true, {S}>: worst {w}, mean {m:.2f}")
    # B fragments: lane L of a block reads ring + 1024 * piece + 16 * L: 16 consecutive lanes = 256 consecutive bytes = all 64 banks once
    print("B-fragment reads: 16 lanes x 16 B consecutive = 64 banks once: 1 cycle per pass by construction")
