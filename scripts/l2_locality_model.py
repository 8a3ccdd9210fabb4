"""LRU model of one XCD L2 for the level-1 gathers of the unit-balanced 3^3 conv and the tile-stationary weight
gradient kernel under alternative mask-sort chunkings / dispatch orders (CPU only; uses the bench batch)."""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
import bench
from oracle import sparse_ref as sr
b = bench.get_batch(seed=0, batch_size=4, voxel_size=0.025)
C = np.asarray(b["sinput0_C"]).astype(np.int64)
print("rows", C.shape)
# neighbour table for 3^3 stride 1
keys = sr.pack_keys(C)
order = np.argsort(keys); sk = keys[order]
offs = sr.region_offsets(3)
print(len(offs))
n = len(C)
nbr = np.full((27, n), -1, np.int64)
for k, o in enumerate(offs):
    q = C.copy(); q[:,1:4] += np.asarray(o, dtype=np.int64)
    qk = sr.pack_keys(q)
    pos = np.searchsorted(sk, qk); pos[pos>=n] = n-1
    hit = sk[pos] == qk
    nbr[k, hit] = order[pos[hit]]
mask = np.zeros(n, np.uint32)
for k in range(27): mask |= ((nbr[k]>=0).astype(np.uint32) << k)
pairs = int((nbr>=0).sum())
print("pairs", pairs, "per row", pairs/n)
np.savez("/tmp/l0map.npz", nbr=nbr, mask=mask)
def redundancy(perm, tm=128):
    m = mask[perm]
    nt = (n + tm - 1)//tm
    pad = np.zeros(nt*tm, np.uint32); pad[:n] = m
    t = np.bitwise_or.reduce(pad.reshape(nt, tm), axis=1)
    pc = np.array([bin(x).count("1") for x in t])
    return pc.sum()*tm/pairs, pc
for sub in (0, 8, 16, 32, 64, 128, 256):
    if sub == 0:
        perm = np.arange(n)
    else:
        chunk = -(-(-(-n//128))//sub)*128
        key = (np.arange(n)//chunk).astype(np.uint64) << np.uint64(27) | mask.astype(np.uint64)
        perm = np.argsort(key, kind="stable")
    r128,_ = redundancy(perm,128); r64,_ = redundancy(perm,64); r16,_=redundancy(perm,16)
    print("chunks %4d  redundancy tile128 %.3f tile64 %.3f group16 %.3f" % (sub, r128, r64, r16))

# ---- conv: unit-balanced launch in R dispatch rounds
from collections import OrderedDict
d = np.load("/tmp/l0map.npz"); nbr = d["nbr"]; mask = d["mask"]
n = nbr.shape[1]
XR = 21845  # rows of one XCD's chunk (1/4 of this cloud ~ 1/8 of the pair tensor)
rows0 = np.arange(0, XR)
def lru_misses(stream, cap):
    c = OrderedDict(); miss = 0
    for r in stream:
        if r in c: c.move_to_end(r)
        else:
            miss += 1; c[r] = 1
            if len(c) > cap: c.popitem(last=False)
    return miss
def conv_stream(sub, R, wgs=96, tm=128):
    # sort inside sub-chunks of the XCD chunk
    chunk = -(-(-(-XR//128))//sub)*128
    key = (rows0//chunk).astype(np.uint64) << np.uint64(27) | mask[rows0].astype(np.uint64)
    perm = rows0[np.argsort(key, kind="stable")]
    nt = -(-XR//tm)
    # units: (tile, k) for k in tile mask
    units = []
    for t in range(nt):
        rr = perm[t*tm:(t+1)*tm]
        tmk = np.bitwise_or.reduce(mask[rr])
        for k in range(27):
            if (tmk >> k) & 1: units.append((t, k))
    U = len(units); G = wgs*R; per = -(-U//G)
    # rounds: WGs [r*wgs, (r+1)*wgs) run concurrently, step-interleaved
    stream = []
    for r in range(R):
        for s in range(per):
            for g in range(r*wgs, (r+1)*wgs):
                u = g*per + s
                if u >= U: continue
                t, k = units[u]
                rr = perm[t*tm:(t+1)*tm]
                v = nbr[k, rr]; stream.append(v[v >= 0])
    return np.concatenate(stream), U
for sub, R in ((1,1),(2,1),(2,2),(4,1),(4,4),(8,8),(8,1)):
    st, U = conv_stream(sub, R)
    for cap in (4000, 6800):
        m = lru_misses(st.tolist(), cap)
        print("sub %d rounds %d units %d accesses %d  cap %d rows: misses %d (%.1f%%)  distinct %d" % (sub, R, U, len(st), cap, m, 100*m/len(st), len(set(st.tolist()))))
