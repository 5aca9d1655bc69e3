"""Which key should a launch dispatch its patches by?  Replays the per-tile lifetimes of profiles/keys_probe.py (one stamps
file per camera position) through list scheduling over the chip's wave slots, for the orders a launch could have had:
    python3 profiles/sim_keys.py gpurun_out/keys [slots]
  place      bottom-up by place (no order)
  own        the view's own patches by their longest tile (what a standing view's history gives)
  stale      the PREVIOUS view's patches by their longest tile, by place (what a camera on the move gets today)
  glass      this view's classification: (reaches a glass primitive, primitives reached) per patch
  content    cost by content: the previous view's longest tile per patch SIGNATURE (the primitives the patch's tiles can reach),
             looked up with this view's signatures
Each with the launch's first round (256 patches) taken from the bottom rows by place, as a launch that classifies at its own
head must (the first round does not wait for the classification), and without that constraint."""
import sys, json, heapq
import numpy as np
base = sys.argv[1]
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
meta = json.load(open(base + "_views.json"))
W, H = meta["width"], meta["height"]
n_patches = (W // 32) * (H // 32)
GLASS = (1 << 0) | (1 << 5)          # demo scene pids: spheres red 0, blue 1 (glass), green 2, white 3; floor quad 4 (glass); triangle 5

def load(k):
    a = np.fromfile("%s_%d.bin" % (base, k), dtype=np.uint64).reshape(-1, 4)
    x = np.fromfile("%s_%d.bin.ext" % (base, k), dtype=np.uint64).reshape(-1, 2)
    life = (a[:, 2].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.0
    tile = x[:, 0].astype(np.int64)
    ok = (tile >= 0) & (tile < n_patches * 16) & (a[:, 2] > 0)
    L = np.full(n_patches * 16, np.nan)
    M = np.zeros(n_patches * 16, dtype=np.uint64)
    L[tile[ok]] = life[ok]
    M[tile[ok]] = x[ok, 1]
    span = (a[a[:, 2] > 0, 2].max() - a[a[:, 0] > 0, 0].min()) / 100.0
    head = life[(tile < 0) | (tile >= n_patches * 16)]
    return L, M, span, head

def sim(lives, dispatch_ns=0.25):
    h = [0.0] * slots
    heapq.heapify(h)
    t_disp = 0.0
    endmax = 0.0
    for Lw in lives:
        t = max(heapq.heappop(h), t_disp)
        t_disp = t + dispatch_ns * 1e-3
        e = t + Lw
        endmax = max(endmax, e)
        heapq.heappush(h, e)
    return endmax

def launch(L, order, static_first):
    """order: patches, first to last; static_first: the bottom 256 patches by place go first whatever the order says"""
    order = list(order)
    if static_first:
        first = list(range(n_patches - 1, n_patches - 1 - slots // 16, -1))
        s = set(first)
        order = first + [p for p in order if p not in s]
    lives = np.concatenate([L[p * 16:(p + 1) * 16][::-1] for p in order])
    return sim(lives)

views = [load(v["k"]) for v in meta["views"]]
print("view  camera            measured | place | own    own+1st | stale  stale+1st | glass+1st | content+1st | prim+1st")
tot = np.zeros(9)
for k, (L, M, span, head) in enumerate(views):
    L = np.where(np.isnan(L), 1.0, L)                  # tiles a tail wave stored: about a sky wave's share
    P = L.reshape(-1, 16)
    Mp = M.reshape(-1, 16)
    place = np.arange(n_patches)[::-1]
    own = np.argsort(-P.max(axis=1), kind="stable")
    Lp, Mprev = views[k - 1][0], views[k - 1][1]
    Lp = np.where(np.isnan(Lp), 1.0, Lp)
    stale = np.argsort(-Lp.reshape(-1, 16).max(axis=1), kind="stable")
    # this view's classification
    known = Mp != np.uint64(0xFFFFFFFFFFFFFFFF)
    m = np.where(known, Mp, np.uint64(0))
    sig = np.bitwise_or.reduce(m, axis=1)
    pop = np.array([bin(int(s)).count("1") for s in sig])
    glass = (sig & np.uint64(GLASS)) != 0
    gkey = np.where(sig == 0, -1, glass * 8 + pop)
    og = np.argsort(-gkey[::-1], kind="stable")
    og = (n_patches - 1 - og)                           # ties: bottom-up
    opop = np.argsort(-np.where(sig == 0, -1, pop)[::-1], kind="stable"); opop = n_patches - 1 - opop
    # cost by content from the previous view
    mprev = np.where(Mprev.reshape(-1, 16) != np.uint64(0xFFFFFFFFFFFFFFFF), Mprev.reshape(-1, 16), np.uint64(0))
    sigp = np.bitwise_or.reduce(mprev, axis=1)
    T = {}
    for s, c in zip(sigp, Lp.reshape(-1, 16).max(axis=1)):
        T[int(s)] = max(T.get(int(s), 0.), c)
    ckey = np.array([(-1. if s == 0 else T.get(int(s), 1e9)) for s in sig])
    oc = np.argsort(-ckey[::-1], kind="stable"); oc = n_patches - 1 - oc
    # cost by primitive: the previous view's mean patch cost (its longest tile) over the patches that can reach the primitive;
    # a patch of this view is as dear as the dearest primitive it can reach
    cs, cn = np.zeros(64), np.zeros(64)
    for s_, c_ in zip(sigp, Lp.reshape(-1, 16).max(axis=1)):
        for b in range(64):
            if (int(s_) >> b) & 1:
                cs[b] += c_; cn[b] += 1
    Cp = np.where(cn > 0, cs / np.maximum(cn, 1), 1e9)
    pkey = np.array([(-1. if s_ == 0 else max(Cp[b] for b in range(64) if (int(s_) >> b) & 1)) for s_ in sig])
    BUCKETS = 16                                        # what a launch would sort by: a few buckets, not the value itself
    top = pkey[pkey < 1e8].max() if (pkey < 1e8).any() else 1.
    pq = np.where(pkey < 0, -1, np.minimum(BUCKETS - 1, np.floor(np.minimum(pkey, top) / top * (BUCKETS - 1))))
    op = n_patches - 1 - np.argsort(-pq[::-1], kind="stable")
    row = [span, launch(L, place, False), launch(L, own, False), launch(L, own, True), launch(L, stale, False), launch(L, stale, True),
           launch(L, og, True), launch(L, oc, True), launch(L, op, True)]
    tot += row
    print("%3d  %-16s  %7.1f  | %5.1f | %5.1f  %5.1f   | %5.1f  %5.1f     | %5.1f     | %5.1f       | %5.1f   sum/slots %.1f, lit tiles %d" % (
        k, meta["views"][k]["camera"], *row, L.sum() / slots, (sig != 0).sum() * 16))
print("mean" + " " * 19 + "  ".join("%5.1f" % v for v in tot / len(views)))

if len(sys.argv) > 3:                                   # a closer look at one view
    k = int(sys.argv[3])
    L, M, span, head = views[k]
    L = np.where(np.isnan(L), 1.0, L)
    P = L.reshape(-1, 16); sig = np.bitwise_or.reduce(M.reshape(-1, 16), axis=1)
    deep = (sig & np.uint64(1)) != 0
    for name, key in (("pid0 first", deep * 2 + (sig != 0)), ("max tile", P.max(axis=1)), ("max tile rounded to 8 us", P.max(axis=1) // 8),
                      ("sum", P.sum(axis=1)), ("pid0, then sum", deep * 1000 + P.sum(axis=1))):
        o = n_patches - 1 - np.argsort(-np.asarray(key, dtype=float)[::-1], kind="stable")
        print("%-28s free %.1f  first round by place %.1f" % (name, launch(L, o, False), launch(L, o, True)))
