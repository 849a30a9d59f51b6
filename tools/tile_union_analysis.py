"""CPU analysis behind profiles/r02_a_sorted_tiles_experiment.txt: how many (tile, offset) products a 16/32/64-row tile has to
execute under different row orders (first-touch, spatial, Morton, neighbour-mask sorts, reduced sort keys), and how the rounds
per workgroup balance under the mask sort.  Uses oracle/ rulebooks on the synthetic KITTI / Waymo clouds (test infrastructure).
usage: python tools/tile_union_analysis.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from vision3d_amd import synth

def stages(cloud, bounds, vs=(0.05,0.05,0.1), max_vox=400000):
    v, co, num = orc.voxelize(cloud, vs, bounds, 5, max_vox)
    b = np.asarray(bounds)
    gx = np.round((b[3:]-b[:3])/np.asarray(vs)).astype(int)  # x,y,z
    shape = [int(gx[2])+1, int(gx[1]), int(gx[0])]
    coords = np.concatenate([np.zeros((co.shape[0],1),np.int32), co],1)
    out = []
    specs = [(3,2,1),(3,2,1),(3,2,[0,1,1]),((3,1,1),(2,1,1),0)]
    for s,(k,st,pd) in enumerate(specs):
        nbr = orc.subm_rulebook(coords, shape, 3)
        out.append((f"subm{s}", coords, nbr))
        oc, nb2, oshape = orc.sparse_rulebook(coords, shape, k, st, pd)
        out.append((f"sp{s}", oc, nb2))
        coords, shape = oc, oshape
    return out

def waste(nbr, order, T):
    n = nbr.shape[0]
    m = (nbr[order] >= 0)
    pad = (-n) % T
    if pad: m = np.concatenate([m, np.zeros((pad, m.shape[1]), bool)])
    tiles = m.reshape(-1, T, m.shape[1])
    union = tiles.any(1).sum()
    pairs = m.sum()
    return union*T/ max(pairs,1), union / (tiles.shape[0]*m.shape[1])

def morton(c):
    def part(x):
        x = x.astype(np.uint64)
        r = np.zeros_like(x)
        for i in range(12): r |= ((x>>np.uint64(i))&np.uint64(1)) << np.uint64(3*i)
        return r
    return part(c[:,3]) | (part(c[:,2])<<np.uint64(1)) | (part(c[:,1])<<np.uint64(2))

def maskkey(nbr):
    m = (nbr>=0).astype(np.uint64)
    w = (np.uint64(1) << np.arange(m.shape[1],dtype=np.uint64))
    return (m*w).sum(1)

for name, cloud, bounds in [("kitti", synth.make_cloud(0), synth.KITTI_BOUNDS), ("waymo", synth.make_waymo_cloud(0), synth.WAYMO_BOUNDS)]:
    for lname, coords, nbr in stages(cloud, bounds):
        n = nbr.shape[0]
        fan = (nbr>=0).sum()/n
        orders = {
          "id": np.arange(n),
          "zyx": np.lexsort((coords[:,3],coords[:,2],coords[:,1])),
          "yxz": np.lexsort((coords[:,1],coords[:,3],coords[:,2])),
          "morton": np.argsort(morton(coords), kind="stable"),
          "mask": np.argsort(maskkey(nbr), kind="stable"),
        }
        # popcount-then-mask
        s = []
        for T in (16,32,64):
            s.append(" ".join(f"{k}:{waste(nbr,o,T)[0]:.2f}/{waste(nbr,o,T)[1]:.2f}" for k,o in orders.items()))
        print(f"{name} {lname} N={n} K={nbr.shape[1]} fan={fan:.1f} dense_waste={nbr.shape[1]/fan:.2f}")
        for T,x in zip((16,32,64),s): print(f"   T={T}: {x}")

print("==== local chunk sort / reduced keys, T=16 and 32")
def chunk_sort(key, C):
    n = key.shape[0]
    order = np.arange(n)
    for s in range(0, n, C):
        seg = order[s:s+C]
        order[s:s+C] = seg[np.argsort(key[seg], kind="stable")]
    return order
for name, cloud, bounds in [("kitti", synth.make_cloud(0), synth.KITTI_BOUNDS), ("waymo", synth.make_waymo_cloud(0), synth.WAYMO_BOUNDS)]:
    for lname, coords, nbr in stages(cloud, bounds):
        if nbr.shape[1] != 27: continue
        key = maskkey(nbr)
        m = nbr >= 0
        # reduced key: 8 in-plane bits + any-above + any-below
        inpl = [9,10,11,12,14,15,16,17]
        rk = sum((m[:,k].astype(np.int64) << i) for i,k in enumerate(inpl)) | (m[:,:9].any(1).astype(np.int64) << 8) | (m[:,18:].any(1).astype(np.int64) << 9)
        res = []
        for C in (256, 1024, 4096):
            o = chunk_sort(key, C)
            res.append(f"chunk{C}:{waste(nbr,o,16)[1]:.2f}/{waste(nbr,o,32)[1]:.2f}")
        o = np.argsort(rk, kind="stable"); res.append(f"rk10:{waste(nbr,o,16)[1]:.2f}/{waste(nbr,o,32)[1]:.2f}")
        o = np.argsort(key >> np.uint64(13), kind="stable"); res.append(f"top14:{waste(nbr,o,16)[1]:.2f}/{waste(nbr,o,32)[1]:.2f}")
        o = np.argsort(key, kind="stable"); res.append(f"full:{waste(nbr,o,16)[1]:.2f}/{waste(nbr,o,32)[1]:.2f}")
        print(name, lname, nbr.shape[0], "ideal:%.2f" % (m.sum()/m.size), " ".join(res))

print("==== reduced keys for strided + subm")
for name, cloud, bounds in [("kitti", synth.make_cloud(0), synth.KITTI_BOUNDS), ("waymo", synth.make_waymo_cloud(0), synth.WAYMO_BOUNDS)]:
    for lname, coords, nbr in stages(cloud, bounds):
        if nbr.shape[1] != 27: continue
        key = maskkey(nbr)
        m = nbr >= 0
        m3 = m.reshape(-1,3,3,3)
        def bits(arr):  # arr (N, nb) bool -> int key, first column = LSB
            return sum((arr[:,i].astype(np.int64) << i) for i in range(arr.shape[1]))
        inpl = m3.any(1).reshape(-1,9)       # OR over kz
        zany = m3.any((2,3))                  # (N,3)
        kA = bits(np.concatenate([inpl, zany],1))     # 12 bits, z-any most significant
        kB = bits(np.concatenate([zany, inpl],1))     # in-plane most significant
        mid = m3[:,1].reshape(-1,9)
        kC = bits(np.concatenate([mid, m3[:,0].any((1,2))[:,None], m3[:,2].any((1,2))[:,None]],1))  # rk10 (with center)
        yany = m3.any((1,3)); xany = m3.any((1,2))
        kD = bits(np.concatenate([xany, yany, zany],1))  # 9 bits separable
        res=[]
        for nm,k in (("A12",kA),("B12",kB),("C11",kC),("D9",kD)):
            o = np.argsort(k, kind="stable"); res.append(f"{nm}:{waste(nbr,o,16)[1]:.2f}/{waste(nbr,o,64)[1]:.2f}")
        o = np.argsort(key, kind="stable"); res.append(f"full:{waste(nbr,o,16)[1]:.2f}/{waste(nbr,o,64)[1]:.2f}")
        # two-level: A12 then full within? (= sort by (kA, key))
        o = np.lexsort((key, kA)); res.append(f"A12+full:{waste(nbr,o,16)[1]:.2f}/{waste(nbr,o,64)[1]:.2f}")
        print(name, lname, nbr.shape[0], "ideal:%.2f" % (m.sum()/m.size), " ".join(res))

print("==== rounds per WG distribution (A12 sort)")
for name, cloud, bounds in [("waymo", synth.make_waymo_cloud(0), synth.WAYMO_BOUNDS), ("kitti", synth.make_cloud(0), synth.KITTI_BOUNDS)]:
    for lname, coords, nbr in stages(cloud, bounds):
        if nbr.shape[1] != 27: continue
        m = nbr >= 0
        m3 = m.reshape(-1,3,3,3)
        def bits(arr): return sum((arr[:,i].astype(np.int64) << i) for i in range(arr.shape[1]))
        kA = bits(np.concatenate([m3.any(1).reshape(-1,9), m3.any((2,3))],1))
        o = np.argsort(kA, kind="stable")
        mm = m[o]
        for rows in (64, 128, 256):
            pad = (-len(mm)) % rows
            x = np.concatenate([mm, np.zeros((pad,27),bool)]).reshape(-1, rows, 27)
            rounds = x.any(1).sum(1)
            # tile-steps per WG (MFMA work)
            t16 = x.reshape(x.shape[0], rows//16, 16, 27).any(2).sum((1,2))
            print(f"{name} {lname} rows/WG={rows}: WGs={len(rounds)} rounds mean={rounds.mean():.1f} max={rounds.max()} p90={np.percentile(rounds,90):.0f}; tile-steps/WG mean={t16.mean():.1f} max={t16.max()}; sum rounds/256={rounds.sum()/256:.1f}")
