"""The workgroup numberings of round 5 (sketchformer_amd/csrc/skf_common.h: skf_xcd_remap, skf_part_major, skf_deal_rank and the two uses in
skf_bf16_attention.hip), restated in Python: every numbering must be a bijection of the grid (each (sample, head[, block]) exactly once),
and the properties the kernels rely on - whole samples per XCD, every shader engine of an XCD (arrival index % 4) drawing the same mix of
sorted ranks, 32 consecutive workgroups of the block-major numbering carrying one block - are checked on the dispatcher model of
tools/dispatch_sim.py (workgroup id -> XCD id % 8, arrival id // 8, engine arrival % 4).  Device code cannot run here; the formulas are short
enough to keep in step by eye, and the GPU tests check that results do not depend on them."""
import itertools


def xcd_remap(orig, nwg):
    xcd, q, r = orig & 7, nwg >> 3, nwg & 7
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (orig >> 3)


def part_major(lid, ngroups, nparts, G=32):
    chunk = lid // (G * nparts)
    j = lid - chunk * G * nparts
    gc = min(G, ngroups - chunk * G)
    w = j // gc
    return chunk * G + j - w * gc, w


def deal_rank(bid, H):
    x, i = bid & 7, bid >> 3
    m = i // H
    return (x + 8 * m) * H + (i - m * H)


def test_xcd_remap_and_part_major_are_bijections():
    for nwg in (8, 24, 1000, 1024, 4096, 37):
        assert sorted(xcd_remap(b, nwg) for b in range(nwg)) == list(range(nwg))
    for ngroups, nparts in ((1024, 4), (1024, 8), (40, 4), (33, 3), (5, 1), (96, 2)):
        seen = sorted(part_major(l, ngroups, nparts) for l in range(ngroups * nparts))
        assert seen == sorted(itertools.product(range(ngroups), range(nparts)))
    # 32 consecutive ids of a full chunk carry one part, part 0 first
    for l0 in range(0, 1024 * 4, 32):
        parts = {part_major(l, 1024, 4)[1] for l in range(l0, l0 + 32)}
        assert len(parts) == 1 and parts.pop() == (l0 // 32) % 4


def test_deal_rank_keeps_samples_on_one_xcd_and_gives_every_engine_the_same_mix():
    for B, H in ((128, 8), (16, 8), (8, 4), (64, 2), (32, 1), (24, 16)):
        n = B * H
        ks = [deal_rank(b, H) for b in range(n)]
        assert sorted(ks) == list(range(n))                       # bijection
        for bid, k in enumerate(ks):
            assert (k // H) % 8 == bid % 8                        # sample rank r runs on XCD r % 8 - with all its heads
        if H % 4 == 0 and B >= 32:
            # ranks of the samples an engine draws: every engine of every XCD sees the SAME number of samples from each eighth of the sorted list
            per_engine = {}
            for bid, k in enumerate(ks):
                per_engine.setdefault((bid % 8, (bid // 8) % 4), []).append(k // H)
            hist = {e: tuple(sum(1 for r in rs if r * 8 // B == o) for o in range(8)) for e, rs in per_engine.items()}
            assert len(set(hist.values())) == 1, hist


def test_bf16_attention_numberings_are_bijections():
    """attn_bf16_q_kernel / attn_bf16_kv_kernel with a sorted sample list (B % 8 == 0) and without."""
    for B, H, nb in ((128, 8, 4), (32, 8, 3), (8, 2, 4)):
        n = B * H * nb
        # forward / dQ pass, list given
        seen = set()
        for bid in range(n):
            i = bid >> 3
            m = i // nb
            r = deal_rank((bid & 7) + 8 * m, H)
            qb = (i - m * nb + m + (m >> 3)) % nb
            seen.add((r, qb))
        assert len(seen) == n and all(0 <= r < B * H and 0 <= q < nb for r, q in seen)
        # forward / dQ pass, no list: block rotated by (b, h)
        seen = set()
        for bid in range(n):
            lid = xcd_remap(bid, n)
            bh = lid // nb
            seen.add((bh, (lid - bh * nb + bh + (bh >> 3)) % nb))
        assert len(seen) == n
        # dK / dV pass, list given: block-major inside an XCD
        seen = set()
        for bid in range(n):
            m, kbk = part_major(bid >> 3, (B * H) >> 3, nb)
            seen.add((deal_rank((bid & 7) + 8 * m, H), kbk))
        assert len(seen) == n and all(0 <= r < B * H and 0 <= k < nb for r, k in seen)
