"""Bank model of the [t][c] LDS image of wgrad16t_kernel (wg_wgrad16t.h): 32 time steps x 128 channels of bf16 in UNPADDED 256-byte
rows; the 16-byte unit (8 channels of one time step) with index ch (0..15) of row t sits at unit position ch ^ F(t),
    F(t) = ((t >> 3) & 1) << 3 | (t & 3) << 1 | ((t >> 2) & 1).
Checked here (MI355X_MICROARCH.md, LDS section: ds_read_b64_tr_b16 is served per 32-lane half on 64 banks of 4 bytes, ds_write_b128 per
group of 8 contiguous lanes on 32 banks):
  * the operand fetch of v_mfma_f32_16x16x32_bf16 through ds_read_b64_tr_b16: the 16-lane group g of a wave reads the 4 x 16 block
    rows 8g + {0..3} (second read: 8g + 4 + {0..3}), columns 16 mb .. 16 mb + 15; lane 4q + p supplies row q, columns 4p .. 4p + 3;
  * the loaders' staging write: lane l of a wave stores the unit (t = l & 31, channel group cg0 + (l >> 5)) -- eight consecutive lanes
    hold eight consecutive time steps of one channel group, i.e. one 128-byte line of the S-plane.
Also prints the same two numbers for the padded 320-byte-row image with the 32-byte rotation that wgrad16s uses (32x32x16 shape).
    python tools/experiments/lds_tr_layout_check.py
"""


def F(t):
    return (((t >> 3) & 1) << 3) | ((t & 3) << 1) | ((t >> 2) & 1)


def off(t, ch):
    return 256 * t + 16 * (ch ^ F(t))


def worst(addrs_by_group, nbanks, width):
    w = 0
    for addrs in addrs_by_group:
        banks = {}
        for a in addrs:
            for d in range(width // 4):
                banks.setdefault(((a // 4) + d) % nbanks, set()).add(a // 4 + d)
        w = max(w, max(len(v) for v in banks.values()))
    return w


def tr_read(mb, second):
    groups = []
    for half in range(2):
        addrs = []
        for l in range(32 * half, 32 * half + 32):
            g, q, p = l >> 4, (l & 15) >> 2, l & 3
            t = 8 * g + q + (4 if second else 0)
            addrs.append(off(t, 2 * mb + (p >> 1)) + 8 * (p & 1))
        groups.append(addrs)
    return groups


def stage_write(cg0):
    groups = []
    for g0 in range(0, 64, 8):
        groups.append([off(l & 31, cg0 + (l >> 5)) for l in range(g0, g0 + 8)])
    return groups


if __name__ == "__main__":
    r = max(worst(tr_read(mb, s), 64, 8) for mb in range(8) for s in (0, 1))
    w = max(worst(stage_write(cg0), 32, 16) for cg0 in range(0, 16, 2))
    print("swizzled 256-byte rows: transposing read %d-way, staging write %d-way (1 = conflict free)" % (r, w))
    assert r == 1 and w == 1
    # data check: every (t, ch) maps to a distinct unit
    assert len({off(t, ch) for t in range(32) for ch in range(16)}) == 512


# ---- round 5: the image LDS-DMA would write for the weight-gradient product (next step for wgrad16t) ----
# global_load_lds_dwordx4 copies 1 KB lane-linear: 64 consecutive time steps of ONE 8-channel group of an S-plane ([c/8][p][8]).  The
# image is therefore [channel group][t 0..63][8 channels], a plane of 1 024 bytes per group -- no swizzle inside a piece, but the planes'
# bases are free.  The transposing read of the 16-channel block mb touches the planes 2 mb and 2 mb + 1 at 8 time steps per 32-lane half:
# conflict free iff the two planes are 64 (mod 128) bytes apart, e.g. a plane stride of 1 088 bytes.
def dma_off(t, cg, stride):
    return stride * cg + 16 * t


def dma_tr_read(mb, second, kstep, stride):
    groups = []
    for half in range(2):
        addrs = []
        for l in range(32 * half, 32 * half + 32):
            g, q, p = l >> 4, (l & 15) >> 2, l & 3
            t = 32 * kstep + 8 * g + q + (4 if second else 0)
            addrs.append(dma_off(t, 2 * mb + (p >> 1), stride) + 8 * (p & 1))
        groups.append(addrs)
    return groups


if __name__ == "__main__":
    for stride in (1024, 1024 + 32, 1024 + 64, 1024 + 128):
        r = max(worst(dma_tr_read(mb, s, ks, stride), 64, 8) for mb in range(16) for s in (0, 1) for ks in (0, 1))
        print("LDS-DMA image [cg][t][8], plane stride %4d bytes: transposing read %d-way" % (stride, r))
    assert max(worst(dma_tr_read(mb, s, ks, 1088), 64, 8) for mb in range(16) for s in (0, 1) for ks in (0, 1)) == 1
