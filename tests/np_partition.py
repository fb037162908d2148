"""Test-side restatement of the partition (numpy / plain Python, round 2's implementation): what\npastix_amd_dist_partition (csrc/partition.cpp) must reproduce rank for rank."""
import numpy as np


def cblk_flops(cblk4, blok4):
    c4, b4 = np.asarray(cblk4, dtype=np.int64), np.asarray(blok4, dtype=np.int64)
    nc = len(c4) - 1
    N = (c4[:-1, 1] - c4[:-1, 0] + 1).astype(np.float64)
    S = c4[:-1, 3].astype(np.float64)
    M = S - N
    fl = N * (((1. / 6.) * N + 0.5) * N + (1. / 3.)) + N * (((1. / 6.) * N) * N - (1. / 6.)) + M * N * (N + 1.)
    owner_of_blok = np.repeat(np.arange(nc), np.diff(c4[:, 2]))
    h = (b4[:, 1] - b4[:, 0] + 1).astype(np.float64)
    offd = b4[:, 3] > 0
    rem = S[owner_of_blok] - b4[:, 3]
    g = 2.0 * rem * h * N[owner_of_blok] * offd
    fl += np.bincount(owner_of_blok, weights=g, minlength=nc)
    return fl


def _etree(c4, b4):
    nc = len(c4) - 1
    nb = np.diff(c4[:, 2])
    parent = np.full(nc, -1, dtype=np.int64)
    has = nb > 1
    parent[has] = b4[c4[:-1, 2][has] + 1, 2]
    return parent


def partition(cblk4, blok4, world, split=6, light=0.05):
    """owner[k] for every cblk by proportional mapping (the idea of PaStiX's blend, splitpart.c:752-1012, on the
    cblk elimination tree): a subtree is given a SET of ranks; the chain of cblks at its top (the split cblks of
    one separator) is dealt over that set, longest first onto the least loaded rank; where the tree branches the
    set is divided among the heavy children in proportion to their work; a subtree with one rank goes to it
    whole.  Light side subtrees (< `light` of their parent's work) go whole to the least loaded rank of the set.
    Fan-in traffic therefore stays inside the rank set of the enclosing subtree: a rank only contributes to
    separators on its own path to the root.  `split` is kept for callers of the earlier interface (unused)."""
    del split
    c4, b4 = np.asarray(cblk4, dtype=np.int64), np.asarray(blok4, dtype=np.int64)
    nc = len(c4) - 1
    owner = np.full(nc, -1, dtype=np.int32)
    if world <= 1:
        owner[:] = 0
        return owner
    fl = cblk_flops(c4, b4)
    parent = _etree(c4, b4)
    sub = fl.copy()
    kids = [[] for _ in range(nc)]
    for k in range(nc):                           # children have smaller indices than their parents
        q = parent[k]
        if q >= 0:
            sub[q] += sub[k]
            kids[q].append(k)
    load = np.zeros(world)
    whole = []                                    # (subtree root, rank): everything below goes to the rank

    def give_whole(root, ranks):
        q = min(ranks, key=lambda r_: load[r_])
        whole.append((root, q))
        load[q] += sub[root]

    def split_ranks(ranks, weights):
        """Divide the rank list among len(weights) <= len(ranks) children, proportionally, at least one each."""
        m, tot = len(ranks), float(sum(weights))
        cnt = [max(1, int(round(m * w_ / tot))) for w_ in weights]
        while sum(cnt) > m:
            i = max(range(len(cnt)), key=lambda j: (cnt[j] > 1, cnt[j] - m * weights[j] / tot))
            cnt[i] -= 1
        while sum(cnt) < m:
            i = max(range(len(cnt)), key=lambda j: m * weights[j] / tot - cnt[j])
            cnt[i] += 1
        out, pos = [], 0
        for c_ in cnt:
            out.append(ranks[pos:pos + c_])
            pos += c_
        return out

    stack = [(r_, list(range(world))) for r_ in range(nc) if parent[r_] < 0]
    if len(stack) > 1:                            # a forest: treat the roots as children of a virtual node
        roots = sorted((r_ for r_, _ in stack), key=lambda r_: -sub[r_])
        stack = []
        heavy = roots[:world]
        for r_, rk in zip(heavy, split_ranks(list(range(world)), [sub[r_] for r_ in heavy])):
            stack.append((r_, rk))
        for r_ in roots[world:]:
            give_whole(r_, list(range(world)))
    while stack:
        node, ranks = stack.pop()
        if len(ranks) == 1:
            whole.append((node, ranks[0]))
            load[ranks[0]] += sub[node]
            continue
        chain = []
        while True:                               # walk down the separator chain to the branching point
            chain.append(node)
            ch = sorted(kids[node], key=lambda c_: -sub[c_])
            heavy = [c_ for c_ in ch if sub[c_] >= light * sub[node]]
            for c_ in ch[len(heavy):]:
                give_whole(c_, ranks)
            if len(heavy) != 1:
                break
            node = heavy[0]
        for k in sorted(chain, key=lambda c_: -fl[c_]):
            q = min(ranks, key=lambda r_: load[r_])
            owner[k] = q
            load[q] += fl[k]
        if not heavy:
            continue
        if len(heavy) > len(ranks):               # more heavy children than ranks: the lightest go whole
            for c_ in heavy[len(ranks):]:
                give_whole(c_, ranks)
            heavy = heavy[:len(ranks)]
        for c_, rk in zip(heavy, split_ranks(ranks, [sub[c_] for c_ in heavy])):
            stack.append((c_, rk))
    for root, q in whole:
        owner[root] = q
    for k in range(nc - 1, -1, -1):               # parents have larger indices than their children
        if owner[k] < 0:
            owner[k] = owner[parent[k]]
    assert (owner >= 0).all()
    return owner


