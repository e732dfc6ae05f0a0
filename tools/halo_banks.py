"""LDS bank check of the halo kernels' fragment reads (conv3d_halo.hip): for both tile shapes, every tap shift and both k-halves, the sixteen 16-byte slots that one
ds_read_b128 lane group requests must be distinct mod 16 (64 banks x 4 B = sixteen 16-byte slots per LDS cycle; groups from MI355X_MICROARCH.md, LDS section)."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def row_of(tw, r):
    g = r >> 2
    if tw == 8:
        return (0x32230110 >> (4 * g)) & 15, (r & 3) + (((0xCC >> g) & 1) << 2)
    return (0x76452310 >> (4 * g)) & 15, r & 3


def check(tw):
    S, SD = (12, 72) if tw == 8 else (6, 40)
    worst = 1
    for blk in range(4):
        for tap in range(27):
            toff = (tap // 9) * SD + ((tap // 3) % 3) * S + tap % 3
            for grp in GROUPS:
                slots = []
                for r in grp:
                    q, x = row_of(tw, r)
                    dz, dy = (blk, q) if tw == 8 else (2 * blk + (q >> 2), q & 3)
                    slots.append((dz * SD + dy * S + x + toff) % 16)
                worst = max(worst, max(slots.count(v) for v in set(slots)))
    # every output position of the tile is covered exactly once
    seen = set()
    for blk in range(4):
        for r in range(32):
            q, x = row_of(tw, r)
            seen.add(((blk, q, x) if tw == 8 else (2 * blk + (q >> 2), q & 3, x)))
    assert len(seen) == 128
    return worst


if __name__ == '__main__':
    for tw in (8, 4):
        print('TW = %d: worst bank multiplicity of a fragment read = %d-way' % (tw, check(tw)))
