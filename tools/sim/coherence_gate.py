"""Gate for a binned swap in the pool kernel: how many (wave-step, object) evaluations of the Cornell
headline frame would wave-level Lipschitz culling remove if a wave's 64 marching rays were drawn BY SPATIAL CELL (origin
octant x direction octant) from the 512 rays a block holds, instead of being unrelated?

CPU simulation in numpy (no GPU, no oracle): the 8-box Cornell scene of raytracingpbr_amd.scene (x10 scale), camera rays at
1920x1080, the relaxed sphere tracing of the v3 variant, diffuse-like secondary rays (n + unit sphere).  Rays are marched in
'waves' of 64 lanes with refill from a queue (a finished lane takes the next ray), every lane keeps the per-object lower
bounds lb_i = last exact |sdf_i| - marched and the upper bound ub = last minimum + moved exactly as nearest_culled does
(rt_device.hpp), and an object is evaluated in a wave-step iff SOME marching lane cannot exclude it.  Compared: queues filled
(a) in random order (= the pool kernel today; the instrumented GPU build measured 1.6 % culled), (b) per block of 512 rays
sorted by cell, (c) globally sorted by cell (upper bound of any binning).  Prints the fraction of evaluations that vanish.
"""
import sys
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from raytracingpbr_amd import cornell_box

rng = np.random.default_rng(0)
sc = cornell_box("v3", aspect=16 / 9)
objs = sc.objects
N = len(objs)
RHO, OMEGA, MIN_DIS, MAX_DIS, HIT_EPS = 0.01, 1.6, 0.05, 2000.0, 0.5 / 1920


def euler(deg):
    rx, ry, rz = np.radians(deg)
    sx, cx, sy, cy, sz, cz = np.sin(rx), np.cos(rx), np.sin(ry), np.cos(ry), np.sin(rz), np.cos(rz)
    Rz = np.array([[cz, sz, 0], [-sz, cz, 0], [0, 0, 1]]); Ry = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]]); Rx = np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]])
    return Rz @ Ry @ Rx


POS = np.array([np.array(o.transform.position, float) * 10 for o in objs])
SCL = np.array([np.array(o.transform.scale, float) * 10 for o in objs])
MAT = np.array([euler(np.array(o.transform.rotation, float)) for o in objs])


def sdf_all(p):                       # p (n,3) -> |sdf| (n,N)
    out = np.empty((len(p), N))
    for i in range(N):
        l = (p - POS[i]) @ MAT[i].T
        q = np.abs(l) - SCL[i]
        out[:, i] = np.abs(np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0) - RHO)
    return out


def march_simple(o, d):               # plain full march: hit flag, hit position, nearest index
    t = np.full(len(o), MIN_DIS); w = np.full(len(o), OMEGA); s = np.zeros(len(o)); ld = np.zeros(len(o))
    alive = np.ones(len(o), bool); hit = np.zeros(len(o), bool); idx = np.zeros(len(o), int); te = t.copy()
    for _ in range(512):
        if not alive.any(): break
        a = np.nonzero(alive)[0]
        D = sdf_all(o[a] + t[a, None] * d[a]); dist = D.min(axis=1); te[a] = t[a]; idx[a] = D.argmin(axis=1)
        fb = (w[a] > 1) & (ld[a] + dist < s[a])
        s_new = np.where(fb, s[a] - w[a] * s[a], w[a] * dist)
        h = ~fb & (dist / np.maximum(t[a], 1e-30) < HIT_EPS)
        w[a] = np.where(fb, 1.0, w[a]); s[a] = s_new; t[a] += s_new; ld[a] = dist
        done = ~fb & (h | (t[a] > MAX_DIS))
        hit[a[h]] = True; alive[a[done]] = False
    return hit, o + te[:, None] * d, idx


# ---- ray population: camera rays -> hits -> diffuse secondary rays (two generations)
W, H, n_cam = 1920, 1080, 60000
cam = sc.camera
lf, la, up = (np.array(v, float) for v in (cam.lookfrom, cam.lookat, cam.vup))
z = (lf - la) / np.linalg.norm(lf - la); x = np.cross(up, z); x /= np.linalg.norm(x); y = np.cross(z, x)
hh = np.tan(np.radians(cam.vfov) / 2); hw = cam.aspect * hh
u, v = rng.random(n_cam), rng.random(n_cam)
dirs = (-z + (2 * u[:, None] - 1) * hw * x + (2 * v[:, None] - 1) * hh * y); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
orig = np.tile(lf, (n_cam, 1))
pop_o, pop_d = [], []
for gen in range(3):
    hit, pos, idx = march_simple(orig, dirs)
    pos, idx = pos[hit], idx[hit]
    e = 0.003
    n = np.stack([sdf_all(pos + np.eye(3)[k] * e)[np.arange(len(pos)), idx] - sdf_all(pos - np.eye(3)[k] * e)[np.arange(len(pos)), idx] for k in range(3)], axis=1)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-12)
    n *= -np.sign((n * dirs[hit]).sum(axis=1, keepdims=True))                 # face the incoming ray
    sph = rng.normal(size=(len(pos), 3)); sph /= np.linalg.norm(sph, axis=1, keepdims=True)
    nd = n + sph; nd /= np.maximum(np.linalg.norm(nd, axis=1, keepdims=True), 1e-9)
    orig, dirs = pos, nd
    pop_o.append(orig); pop_d.append(dirs)
PO, PD = np.concatenate(pop_o), np.concatenate(pop_d)
print(f"secondary rays: {len(PO)}")
centre = np.array([0.0, 0.0, 0.0])
cell = ((PO[:, 0] > centre[0]) * 1 + (PO[:, 1] > centre[1]) * 2 + (PO[:, 2] > centre[2]) * 4) * 8 + (PD[:, 0] > 0) * 1 + (PD[:, 1] > 0) * 2 + (PD[:, 2] > 0) * 4


def culled_fraction(order, n_waves=160):
    """march waves of 64 lanes with refill from `order` (ray indices); returns evaluated / possible object evaluations"""
    ev = tot = 0
    q = 0
    for wv in range(n_waves):
        lane_ray = order[q:q + 64]; q += 64
        if len(lane_ray) < 64: break
        o = PO[lane_ray].copy(); d = PD[lane_ray].copy()
        t = np.full(64, MIN_DIS); w = np.full(64, OMEGA); s = np.zeros(64); ld = np.zeros(64)
        lb = np.full((64, N), -1.0); ub = np.full(64, 3e38)
        budget = 192                                        # rays this wave works off (refills included)
        for step in range(4000):
            D = sdf_all(o + t[:, None] * d)
            eps = 1.9e-6 * (np.abs(t) + 400.0)
            need = lb <= (ub + eps)[:, None]                   # (64, N): the lane cannot exclude the object
            need_wave = need.any(axis=0)
            ev += int(need_wave.sum()); tot += N
            # exact step (the culled objects are never the nearest: use the full evaluation)
            dist = D.min(axis=1)
            lb = np.where(need_wave[None, :], D - eps[:, None], lb)
            fb = (w > 1) & (ld + dist < s)
            s_new = np.where(fb, s - w * s, w * dist)
            h = ~fb & (dist / t < HIT_EPS)
            w = np.where(fb, 1.0, w); s = s_new; t_before = t.copy(); t = t + s_new; ld = dist
            moved = np.abs(t - t_before) * 1.000001
            ub = dist + moved; lb = lb - moved[:, None]
            done = ~fb & (h | (t > MAX_DIS))
            k = int(done.sum())
            if k:
                take = min(k, budget, len(order) - q)
                if take < k: break                            # queue exhausted: end of this wave's measurement
                new = order[q:q + take]; q += take; budget -= take
                di = np.nonzero(done)[0]
                o[di] = PO[new]; d[di] = PD[new]; t[di] = MIN_DIS; w[di] = OMEGA; s[di] = 0; ld[di] = 0; lb[di] = -1.0; ub[di] = 3e38
            if budget <= 0: break
    return ev / tot


n = len(PO)
perm = rng.permutation(n)
res = {"random (today)": culled_fraction(perm)}
blk = perm.copy()
for b in range(0, n - 511, 512):
    seg = blk[b:b + 512]; blk[b:b + 512] = seg[np.argsort(cell[seg], kind="stable")]
res["binned inside blocks of 512 rays (64 cells)"] = culled_fraction(blk)
res["globally sorted by cell (upper bound)"] = culled_fraction(perm[np.argsort(cell[perm], kind="stable")])
for k, v in res.items():
    print(f"{k}: {100 * (1 - v):.1f} % of the (wave-step, object) evaluations vanish ({v * N:.2f} of {N} objects evaluated per wave-step)")
