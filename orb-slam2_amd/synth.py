"""Seeded synthetic inputs shared by the tests, the smoke check and bench.py.

Images: filled axis-aligned rectangles and discs of random size 4..40 px and random grey on a
black background, plus integer noise uniform[-6, 6].  The shapes give thousands of FAST corners
per pyramid level with heavy score ties, which stresses the tie semantics of the per-cell and
per-level keypoint selection.  Descriptors: uniform random 256-bit rows, a quarter of the
queries being noisy copies of train rows and 5 % of the train rows exact duplicates, so that the
best/second-best tie rule and the ratio test are exercised.
"""
import numpy as np


def synth_image(width, height, frame_idx=0):
    """One u8 grayscale frame, fully determined by (width, height, frame_idx)."""
    rng = np.random.default_rng(seed=1000 + frame_idx)
    img = np.zeros((height, width), dtype=np.int16)
    area = width * height / 307200.0
    n_rect = int(round(400 * area))
    n_disc = int(round(200 * area))
    rs = rng.integers(4, 41, size=(n_rect, 2))
    rx = rng.integers(0, width, size=n_rect)
    ry = rng.integers(0, height, size=n_rect)
    rg = rng.integers(0, 256, size=n_rect)
    ds = rng.integers(4, 41, size=n_disc)
    dx = rng.integers(0, width, size=n_disc)
    dy = rng.integers(0, height, size=n_disc)
    dg = rng.integers(0, 256, size=n_disc)
    for i in range(n_rect):
        img[ry[i]:ry[i] + rs[i, 1], rx[i]:rx[i] + rs[i, 0]] = rg[i]
    for i in range(n_disc):
        r = ds[i] / 2.0
        y0, y1 = max(0, int(dy[i] - r)), min(height, int(dy[i] + r) + 1)
        x0, x1 = max(0, int(dx[i] - r)), min(width, int(dx[i] + r) + 1)
        if y1 <= y0 or x1 <= x0:
            continue
        yy, xx = np.ogrid[y0:y1, x0:x1]
        m = (yy - dy[i]) ** 2 + (xx - dx[i]) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = dg[i]
    img += rng.integers(-6, 7, size=img.shape, dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_batch(width, height, batch, first_idx=0):
    return np.stack([synth_image(width, height, first_idx + i) for i in range(batch)])


def synth_descriptors(nq, nt, seed=0):
    """(queries[nq,32], train[nt,32]) u8 with planted near-duplicates and exact duplicates."""
    rng = np.random.default_rng(seed)
    train = rng.integers(0, 256, size=(nt, 32), dtype=np.uint8)
    query = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
    if nt > 1:
        ndup = max(1, nt // 20)
        dst = rng.integers(0, nt, size=ndup)
        src = rng.integers(0, nt, size=ndup)
        train[dst] = train[src]
    if nt > 0 and nq > 0:
        ncopy = nq // 4
        qi = rng.choice(nq, size=ncopy, replace=False)
        ti = rng.integers(0, nt, size=ncopy)
        query[qi] = train[ti]
        nflip = rng.integers(0, 41, size=ncopy)
        for k in range(ncopy):
            bits = rng.integers(0, 256, size=nflip[k])
            for b in bits:
                query[qi[k], b >> 3] ^= np.uint8(1 << (b & 7))
    return query, train


def synth_vocabulary(k, L, seed=0, tie_share=0.15, stop_share=0.05):
    """A synthetic vocabulary tree with branching factor k and depth L, nodes numbered the way the
    reference builds and stores them (a node's children are consecutive and follow it), random 256-bit
    node descriptors, idf-like leaf weights.  A share of sibling groups repeat a descriptor (distance
    ties -> the first child wins) and a share of words are stopped (weight 0).
    Returns (parent int32[n], desc uint8[n, 32], weight float64[n])."""
    rng = np.random.default_rng(seed)
    parent, depth = [0], [0]
    stack = [0]
    while stack:  # depth-first creation, children of a node created together (HKmeansStep)
        p = stack.pop()
        if depth[p] == L:
            continue
        kids = []
        for _ in range(k):
            parent.append(p)
            depth.append(depth[p] + 1)
            kids.append(len(parent) - 1)
        stack.extend(reversed(kids))
    n = len(parent)
    parent = np.array(parent, np.int32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    # children resemble their parent a little, as in a k-means tree
    for i in range(1, n):
        keep = rng.integers(0, 256, 32, dtype=np.uint8) & rng.integers(0, 256, 32, dtype=np.uint8)
        desc[i] = (desc[parent[i]] & ~keep) | (desc[i] & keep)
    for i in range(2, n):
        if parent[i] == parent[i - 1] and rng.random() < tie_share:
            desc[i] = desc[i - 1]
    weight = np.zeros(n, np.float64)
    leaf = np.array(depth) == L
    weight[leaf] = rng.uniform(0.5, 9.0, int(leaf.sum()))
    weight[leaf & (rng.random(n) < stop_share)] = 0.0
    return parent, desc, weight


def synth_vocabulary_large(k, L, seed=0, tie_share=0.05, stop_share=0.05):
    """synth_vocabulary for trees of the size of the ORB vocabulary (k = 10, L = 6: 1,111,111 nodes), generated level
    by level with array operations; same numbering (a node's children are consecutive and are created when a
    depth-first walk reaches the node), same make-up (children resemble their parent, a share of siblings repeat
    the descriptor of the sibling before them, a share of words are stopped), its own random stream.
    Returns (parent int32[n], desc uint8[n, 32], weight float64[n])."""
    rng = np.random.default_rng(seed)
    # size of a complete subtree whose root sits at depth d
    sub = [0] * (L + 2)
    for d in range(L, -1, -1):
        sub[d] = 1 + (k * sub[d + 1] if d < L else 0)
    n = sub[0]
    # Children of a node are numbered when the depth-first walk visits the node: the walk visits the internal nodes in
    # preorder, and the j-th internal node visited owns ids 1 + j k .. k + j k.  Preorder rank among internal nodes of
    # the c-th child (depth d + 1) of a node with rank r: r + 1 + c * (internal nodes of a subtree rooted at depth d + 1).
    internal = [0] * (L + 2)
    for d in range(L - 1, -1, -1):
        internal[d] = 1 + k * internal[d + 1]
    parent = np.zeros(n, np.int32)
    depth = np.zeros(n, np.int8)
    ids = np.array([0], np.int64)    # node ids of the current level, in walk order
    rank = np.array([0], np.int64)   # their preorder ranks among internal nodes
    levels = [ids]
    for d in range(L):
        child_ids = (1 + rank[:, None] * k + np.arange(k)[None, :]).reshape(-1)
        parent[child_ids] = np.repeat(ids, k)
        depth[child_ids] = d + 1
        child_rank = (rank[:, None] + 1 + np.arange(k)[None, :] * internal[d + 1]).reshape(-1)
        ids, rank = child_ids, child_rank
        levels.append(ids)
    desc = np.zeros((n, 32), np.uint8)
    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    for d in range(1, L + 1):
        idl = levels[d]
        own = rng.integers(0, 256, (len(idl), 32), dtype=np.uint8)
        keep = rng.integers(0, 256, (len(idl), 32), dtype=np.uint8) & rng.integers(0, 256, (len(idl), 32), dtype=np.uint8)
        new = (desc[parent[idl]] & ~keep) | (own & keep)
        # a tied node repeats the sibling before it (chains allowed): take the descriptor of the last untied node of its run
        tie = rng.random(len(idl)) < tie_share
        tie[::k] = False  # a first child has no sibling before it
        src = np.maximum.accumulate(np.where(tie, 0, np.arange(len(idl))))
        desc[idl] = new[src]
    weight = np.zeros(n, np.float64)
    leaves = levels[L]
    w = rng.uniform(0.5, 9.0, len(leaves))
    w[rng.random(len(leaves)) < stop_share] = 0.0
    weight[leaves] = w
    return parent, desc, weight
