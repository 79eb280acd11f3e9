"""Index maps of csrc/rmhmc_metric_mfma.hip restated in Python (CPU): the thread -> work-item enumerations the matrix-core metric
kernel relies on cover their index sets exactly once for every size the kernel accepts.  The kernel's results are checked on the
GPU (tests/test_gpu_rmhmc.py); these tests pin the COMBINATORICS, which is what a later edit of the maps can silently break for a
D or a tile count the GPU cases do not happen to use."""
import itertools

import pytest

MT = 1024


def pair_items(D):
    """ph_refine_E / ph_refine_E2: the off-diagonal elements in pairs (i, j), i < j - rows r and D - 1 - r share D - 1 slots;
    thread (c = tid & 127, r = (tid >> 7) + 8 p), p < 7."""
    H = (D + 1) >> 1
    out = []
    for tid in range(MT):
        c = tid & 127
        for p8 in range(7):
            r = (tid >> 7) + 8 * p8
            if r >= H or c >= D - 1:
                continue
            n1 = D - 1 - r
            if c < n1:
                i, j = r, r + 1 + c
            else:
                i, j = D - 1 - r, D - r + (c - n1)
                if i == r:
                    continue
            out.append((i, j))
    return out


@pytest.mark.parametrize("D", list(range(1, 113)))
def test_pair_enumeration_covers_the_strict_upper_triangle_once(D):
    got = pair_items(D)
    want = list(itertools.combinations(range(D), 2))
    assert len(got) == len(want) and set(got) == set(want)


def sym_items(nt):
    """sym_item(): the upper block triangle of a symmetric product as 1 x 2 tile pairs of a tile row, then the odd tile that ends a
    row of odd length - one item per wave (16 waves)."""
    npairs = sum((nt - I) >> 1 for I in range(nt))
    items = []
    for wave in range(16):
        w, I = wave, 0
        if w < npairs:
            while True:
                pr = (nt - I) >> 1
                if w < pr:
                    break
                w -= pr
                I += 1
            items.append((I, I + 2 * w, True))
        else:
            w -= npairs
            while I < nt:
                if (nt - I) & 1:
                    if w == 0:
                        break
                    w -= 1
                I += 1
            if I < nt:
                items.append((I, nt - 1, False))
    return items


@pytest.mark.parametrize("nt", range(1, 8))
def test_symmetric_tile_items_cover_the_upper_block_triangle_once(nt):
    tiles = []
    for I, J, two in sym_items(nt):
        tiles.append((I, J))
        if two:
            tiles.append((I, J + 1))
    want = [(I, J) for I in range(nt) for J in range(I, nt)]
    assert sorted(tiles) == want
    # nt = 7 (D = 97 .. 112): every SIMD (wave & 3) gets three pairs and a single = 7 tiles
    if nt == 7:
        per_simd = [0] * 4
        for wave, (I, J, two) in enumerate(sym_items(nt)):
            per_simd[wave & 3] += 2 if two else 1
        assert per_simd == [7, 7, 7, 7]


def full_items(nt):
    """lds_gemm, full product: 2 x 2 macro tiles, the 1 x 2 / 2 x 1 edges of an odd nt, the corner - dealt by size in boustrophedon
    order over the four SIMDs."""
    nf, odd = nt >> 1, nt & 1
    out = []
    for wave in range(16):
        s, q = wave & 3, wave >> 2
        k = 4 * q + 3 - s if (q & 1) else 4 * q + s
        nfull = nf * nf
        if k < nfull:
            r, c = divmod(k, nf)
        else:
            k -= nfull
            if not odd or k > 2 * nf:
                continue
            if k < nf:
                r, c = k, nf
            elif k < 2 * nf:
                r, c = nf, k - nf
            else:
                r, c = nf, nf
        I0, J0 = 2 * r, 2 * c
        for x in range(2):
            for y in range(2):
                if I0 + x < nt and J0 + y < nt:
                    out.append((I0 + x, J0 + y))
    return out


@pytest.mark.parametrize("nt", range(1, 8))
def test_full_product_macro_tiles_cover_every_tile_once(nt):
    assert sorted(full_items(nt)) == [(I, J) for I in range(nt) for J in range(nt)]


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 16, 17, 37, 64, 99, 100, 101, 111, 112])
def test_chunked_contraction_indices_are_a_permutation(D):
    """gemm_macro: k4 = ceil(D / 4) steps; full chunks of four steps take index 16 c + 4 lk + u at step u of lane group lk, the
    steps beyond the last full chunk 4 ks + lk: together every index below 4 k4 exactly once."""
    k4 = (D + 3) // 4
    nchunk = k4 >> 2
    seen = []
    for c in range(nchunk):
        for u in range(4):
            for lk in range(4):
                seen.append(16 * c + 4 * lk + u)
    for ks in range(4 * nchunk, k4):
        for lk in range(4):
            seen.append(4 * ks + lk)
    assert sorted(seen) == list(range(4 * k4))
