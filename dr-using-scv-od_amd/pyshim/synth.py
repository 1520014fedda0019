"""Seeded synthetic LiDAR scans shaped like the datasets BASELINE.json names.

No real SemanticKITTI / ParkingLot data exists in this environment, so tests and bench.py use
ray-cast scenes scanned by a spinning multi-beam sensor that drives forward 1 m per scan: a tilted noisy
ground plane, parked cars, building facades, fences, poles / trunks, small clutter, POROUS vegetation
(tree canopies and bushes: a ray crossing one returns from a random depth inside it, so the returns fill
a volume the way leaves do) and a few moving boxes.  Everything is a pure function of (kind, seq, scan index).

kinds:  K64    64 beams (+2.0 .. -24.8 deg), 2083 columns, ~10 % drop-outs -> ~120 k returns, street scene with
               vegetation / fences / clutter: ~0.35 N points survive Patchwork + the range/FOV filter and occupy
               ~10 k curved voxels of the semantickitti.yaml grid (N_a 0.3-0.4 N, V 8-15 k)
        K64S   the sparse scene of round 1 (cars, 6 walls and 10 poles per 100 m, no vegetation): ~1.7 k voxels
        OS128  128 beams (+-22.5 deg), 2048 columns, no drop-outs -> 262 144 rays; the K64 street between tall
               facades with decks above the road, so that the sky-facing beams return too (~260 k returns)
        PARK   64 beams (+-16.6 deg), 1024 columns (config/parkinglot.yaml geometry), sparse scene
Labels follow SemanticKITTI: 40 ground, 50 building, 51 fence, 70 vegetation, 71 trunk/pole, 99 other object,
10 static car, 252 moving car.
"""
import math

import torch

SEQ_LEN = {0: 4541, 1: 1101, 2: 4661, 3: 801, 4: 271, 5: 2761, 6: 1101, 7: 1101, 8: 4071, 9: 1591, 10: 1201}

KINDS = {
    "K64": dict(beams=64, el_hi=2.0, el_lo=-24.8, cols=2083, drop=0.10, height=1.73, scene="street"),
    "K64S": dict(beams=64, el_hi=2.0, el_lo=-24.8, cols=2083, drop=0.10, height=1.73, scene="sparse"),
    "OS128": dict(beams=128, el_hi=22.5, el_lo=-22.5, cols=2048, drop=0.0, height=1.73, scene="canyon"),
    "PARK": dict(beams=64, el_hi=16.6, el_lo=-16.6, cols=1024, drop=0.05, height=1.83, scene="sparse"),
}


def pose_of(idx):
    """(x, y, z, roll, pitch, yaw) of scan idx: 1 m/scan forward, gentle yaw weave."""
    yaw = math.radians(0.2) * math.sin(idx / 50.0)
    return (float(idx) * 1.0, 0.0, 0.0, 0.0, 0.0, yaw)


def _segment_objects(seq, seg, kind):
    """Static objects of the 100 m road segment `seg` (world frame).  The first draws are those of the round-1 sparse
    scene, so K64S / PARK scans are unchanged; the street / canyon scenes append to them."""
    scene = KINDS[kind]["scene"]
    g = torch.Generator().manual_seed(20241026 + 1000 * seq + 7919 * (seg + 1000) + (0 if kind != "PARK" else 13))
    x0 = 100.0 * seg
    u = lambda n, lo, hi: lo + (hi - lo) * torch.rand(n, generator=g)
    side = lambda n, lo, hi: torch.where(torch.rand(n, generator=g) < 0.5, u(n, lo, hi), u(n, -hi, -lo))
    q = lambda n: torch.round(torch.rand(n, generator=g) * 100) / 100 * 255
    n_car, n_wall, n_cyl = 14, 6, 10
    cars_c = torch.stack([x0 + u(n_car, 0, 100), torch.where(torch.rand(n_car, generator=g) < 0.5, u(n_car, 3.5, 9), u(n_car, -9, -3.5))], 1)
    swap = torch.rand(n_car, generator=g) < 0.3
    car_dx = torch.where(swap, torch.full((n_car,), 0.9), torch.full((n_car,), 2.1))
    car_dy = torch.where(swap, torch.full((n_car,), 2.1), torch.full((n_car,), 0.9))
    walls_c = torch.stack([x0 + u(n_wall, 0, 100), torch.where(torch.rand(n_wall, generator=g) < 0.5, u(n_wall, 11, 25), u(n_wall, -25, -11))], 1)
    cyl_c = torch.stack([x0 + u(n_cyl, 0, 100), torch.where(torch.rand(n_cyl, generator=g) < 0.5, u(n_cyl, 4, 20), u(n_cyl, -20, -4))], 1)
    cyl_r = u(n_cyl, 0.2, 0.5)
    cyl_h = u(n_cyl, 3.0, 8.0)
    o = dict(cars_c=cars_c, car_dx=car_dx, car_dy=car_dy, car_i=q(n_car), walls_c=walls_c, wall_i=q(n_wall),
             wall_hx=torch.full((n_wall,), 10.0), wall_hy=torch.full((n_wall,), 0.15), wall_top=torch.full((n_wall,), 3.0),
             cyl_c=cyl_c, cyl_r=cyl_r, cyl_h=cyl_h, cyl_i=q(n_cyl))
    if scene == "sparse":
        return o
    # ---- street: more facades, fences, clutter, porous vegetation ----
    n_w2 = 8
    w2_c = torch.stack([x0 + u(n_w2, 0, 100), side(n_w2, 10, 22)], 1)
    o["walls_c"] = torch.cat([walls_c, w2_c])
    o["wall_i"] = torch.cat([o["wall_i"], q(n_w2)])
    o["wall_hx"] = torch.cat([o["wall_hx"], u(n_w2, 4.0, 9.0)])
    o["wall_hy"] = torch.cat([o["wall_hy"], u(n_w2, 0.15, 3.0)])
    o["wall_top"] = torch.cat([o["wall_top"], u(n_w2, 3.0, 9.0)])
    n_f = 5
    o["fence_c"] = torch.stack([x0 + u(n_f, 0, 100), side(n_f, 5.5, 10)], 1)
    o["fence_hx"] = u(n_f, 4.0, 9.0)
    o["fence_i"] = q(n_f)
    n_b = 12
    o["clutter_c"] = torch.stack([x0 + u(n_b, 0, 100), side(n_b, 5.0, 12)], 1)
    o["clutter_h"] = u(n_b, 0.25, 0.5)
    o["clutter_top"] = u(n_b, 0.6, 1.6)
    o["clutter_i"] = q(n_b)
    # extra trunks; canopies sit on the first n_can of all trunks
    n_t2 = 8
    t2_c = torch.stack([x0 + u(n_t2, 0, 100), side(n_t2, 4.8, 16)], 1)
    o["cyl_c"] = torch.cat([cyl_c, t2_c])
    o["cyl_r"] = torch.cat([cyl_r, u(n_t2, 0.12, 0.3)])
    o["cyl_h"] = torch.cat([cyl_h, u(n_t2, 2.5, 5.0)])
    o["cyl_i"] = torch.cat([o["cyl_i"], q(n_t2)])
    n_can = 14
    can_r = u(n_can, 1.4, 3.0)
    o["can_c"] = torch.cat([o["cyl_c"][:n_can], (0.45 * o["cyl_h"][:n_can] + can_r)[:, None]], 1)  # z above the ground
    o["can_r"] = can_r
    o["can_i"] = q(n_can)
    n_bush = 16
    bush_r = u(n_bush, 0.5, 1.4)
    by = side(n_bush, 4.6, 13)  # inner edge: clear of the movers' lanes (+-3.5 m, half width 0.9 m)
    o["bush_c"] = torch.cat([torch.stack([x0 + u(n_bush, 0, 100), by + torch.sign(by) * bush_r], 1), (0.5 * bush_r)[:, None]], 1)
    o["bush_r"] = bush_r
    o["bush_i"] = q(n_bush)
    # hedgerows / tree rows: long porous boxes on both sides of the road, what fills the curved voxels of a real street
    n_h = 28
    hy_in = torch.cat([side(n_h // 2, 4.6, 9.0), side(n_h // 2, 11.0, 20.0)])  # inner edge of the row
    o["hedge_hx"] = u(n_h, 3.0, 9.0)
    o["hedge_hy"] = torch.cat([u(n_h // 2, 0.8, 2.0), u(n_h // 2, 1.5, 4.0)])
    o["hedge_c"] = torch.stack([x0 + u(n_h, 0, 100), hy_in + torch.sign(hy_in) * o["hedge_hy"]], 1)
    o["hedge_top"] = torch.cat([u(n_h // 2, 1.8, 4.5), u(n_h // 2, 3.0, 7.0)])
    o["hedge_i"] = q(n_h)
    # parked cars stand clear of the driving lanes of the movers (+-3.5 m)
    o["cars_c"] = torch.stack([cars_c[:, 0], cars_c[:, 1] + (1.3 + car_dy) * torch.sign(cars_c[:, 1])], 1)
    if scene == "canyon":
        # tall continuous facades on both sides + decks above the road: the sky-facing beams of a +-22.5 deg sensor return
        n_fc = 10
        fx = x0 + 10.0 + 20.0 * torch.arange(5, dtype=torch.float32).repeat(2)
        fy = torch.cat([torch.full((5,), 27.0), torch.full((5,), -27.0)])
        o["walls_c"] = torch.cat([o["walls_c"], torch.stack([fx, fy], 1)])
        o["wall_i"] = torch.cat([o["wall_i"], q(n_fc)])
        o["wall_hx"] = torch.cat([o["wall_hx"], torch.full((n_fc,), 10.0)])
        o["wall_hy"] = torch.cat([o["wall_hy"], torch.full((n_fc,), 0.5)])
        o["wall_top"] = torch.cat([o["wall_top"], torch.full((n_fc,), 60.0)])
        o["deck_c"] = torch.tensor([[x0 + 44.0, 0.0]])  # covers x0 .. x0+88 of every 100 m, full width
        o["deck_i"] = q(1)
    return o


def _movers(seq, idx, kind):
    g = torch.Generator().manual_seed(424242 + 1000 * seq + (0 if kind != "PARK" else 17))
    n = 6
    speed = (5 + 10 * torch.rand(n, generator=g)) if kind != "PARK" else (1 + 2 * torch.rand(n, generator=g))
    lane_y = 1.8 if KINDS[kind]["scene"] == "sparse" else 3.5
    lane = torch.where(torch.rand(n, generator=g) < 0.5, torch.full((n,), lane_y), torch.full((n,), -lane_y))
    phase = torch.rand(n, generator=g) * 120.0
    # movers cycle through a window that travels with the sensor so every scan sees some
    t = idx * 0.1
    rel = ((phase + (speed - 10.0) * t) % 120.0) - 60.0
    cx = float(idx) + rel
    i = torch.round(torch.rand(n, generator=g) * 100) / 100 * 255
    return torch.stack([cx, lane], 1), i


def make_scan(seq, idx, kind="K64", device="cpu", with_labels=True):
    """Returns (xyzi float32 [n,4] in the sensor frame, labels int32 [n], pose tuple)."""
    K = KINDS[kind]
    dev = torch.device(device)
    h = K["height"]
    pose = pose_of(idx)
    sx, yaw = pose[0], pose[5]
    seed = 20241026 + 1000 * seq + idx
    g = torch.Generator(device=dev).manual_seed(seed)
    el = torch.linspace(math.radians(K["el_hi"]), math.radians(K["el_lo"]), K["beams"], device=dev)
    az = torch.arange(K["cols"], device=dev, dtype=torch.float32) * (2 * math.pi / K["cols"])
    ce, se = torch.cos(el)[:, None], torch.sin(el)[:, None]
    dxs = (ce * torch.cos(az)[None, :]).reshape(-1)
    dys = (ce * torch.sin(az)[None, :]).reshape(-1)
    dzs = (se * torch.ones_like(az)[None, :]).reshape(-1)
    # world-frame ray (sensor yaw only), origin (sx, 0, 0)
    cy, syw = math.cos(yaw), math.sin(yaw)
    dx = cy * dxs - syw * dys
    dy = syw * dxs + cy * dys
    dz = dzs
    R = dx.numel()
    INF = 1.0e9
    best_t = torch.full((R,), INF, device=dev)
    best_i = torch.zeros(R, device=dev)
    best_l = torch.zeros(R, dtype=torch.int32, device=dev)

    def take(t, inten, label):
        nonlocal best_t, best_i, best_l
        m = t < best_t
        best_t = torch.where(m, t, best_t)
        best_i = torch.where(m, inten, best_i)
        best_l = torch.where(m, torch.full_like(best_l, label), best_l)

    # ground plane z = -h + a*(x - sx) + b*y  (tilt seeded per sequence)
    gs = torch.Generator().manual_seed(99 + seq)
    a, b = (0.02 * torch.randn(2, generator=gs)).tolist()
    den = dz - a * dx - b * dy
    tg = torch.where(den < -1e-6, (-h) / den, torch.full_like(den, INF))
    take(tg, torch.full_like(tg, 0.25 * 255), 40)

    def boxes(c, hx, hy, z0, z1, inten, label):
        # c [m,2] world centres, half sizes hx, hy [m]; z0 / z1 scalars or [m]; slab test, origin (sx,0,0)
        if c.shape[0] == 0:
            return
        c = c.to(dev)
        hx, hy, inten = hx.to(dev), hy.to(dev), inten.to(dev)
        ox = c[:, 0][None, :] - sx
        oy = c[:, 1][None, :]
        idx_ = 1.0 / torch.where(dx.abs() < 1e-9, torch.full_like(dx, 1e-9), dx)[:, None]
        idy = 1.0 / torch.where(dy.abs() < 1e-9, torch.full_like(dy, 1e-9), dy)[:, None]
        idz = 1.0 / torch.where(dz.abs() < 1e-9, torch.full_like(dz, 1e-9), dz)[:, None]
        tx1, tx2 = (ox - hx[None, :]) * idx_, (ox + hx[None, :]) * idx_
        ty1, ty2 = (oy - hy[None, :]) * idy, (oy + hy[None, :]) * idy
        if torch.is_tensor(z0):
            z0 = z0.to(dev)[None, :]
        if torch.is_tensor(z1):
            z1 = z1.to(dev)[None, :]
        tz1, tz2 = (z0 * idz).expand_as(tx1), (z1 * idz).expand_as(tx1)
        tmin = torch.maximum(torch.maximum(torch.minimum(tx1, tx2), torch.minimum(ty1, ty2)), torch.minimum(tz1, tz2))
        tmax = torch.minimum(torch.minimum(torch.maximum(tx1, tx2), torch.maximum(ty1, ty2)), torch.maximum(tz1, tz2))
        hit = (tmax >= tmin) & (tmin > 0.5)
        t = torch.where(hit, tmin, torch.full_like(tmin, INF))
        tb, jb = t.min(dim=1)
        take(tb, inten[jb], label)

    def cylinders(c, r, hh, inten, label):
        c, r, hh, inten = c.to(dev), r.to(dev), hh.to(dev), inten.to(dev)
        ox = (sx - c[:, 0])[None, :]
        oy = (0.0 - c[:, 1])[None, :]
        A = (dx * dx + dy * dy)[:, None]
        Bq = 2 * (ox * dx[:, None] + oy * dy[:, None])
        Cq = ox * ox + oy * oy - (r * r)[None, :]
        disc = Bq * Bq - 4 * A * Cq
        sq = torch.sqrt(torch.clamp(disc, min=0))
        t = (-Bq - sq) / (2 * A + 1e-12)
        z = t * dz[:, None]
        hit = (disc > 0) & (t > 0.5) & (z >= -h) & (z <= (-h + hh)[None, :])
        t = torch.where(hit, t, torch.full_like(t, INF))
        tb, jb = t.min(dim=1)
        take(tb, inten[jb], label)

    def porous(c, r, inten, label, kappa):
        # spheres of foliage: centre c [m,3] (z above the ground), radius r; a ray with chord L inside returns with
        # probability 1 - exp(-kappa L), from a depth drawn from the truncated exponential (leaves, not a surface)
        c, r, inten = c.to(dev), r.to(dev), inten.to(dev)
        ox = (sx - c[:, 0])[None, :]
        oy = (0.0 - c[:, 1])[None, :]
        oz = (0.0 - (c[:, 2] - h))[None, :]
        Bh = ox * dx[:, None] + oy * dy[:, None] + oz * dz[:, None]  # rays are unit vectors
        Cq = ox * ox + oy * oy + oz * oz - (r * r)[None, :]
        disc = Bh * Bh - Cq
        sq = torch.sqrt(torch.clamp(disc, min=0))
        t0 = torch.clamp(-Bh - sq, min=0.5)
        t1 = -Bh + sq
        L = torch.clamp(t1 - t0, min=0)
        u1 = torch.rand(L.shape, generator=g, device=dev)
        p = 1.0 - torch.exp(-kappa * L)
        hit = (disc > 0) & (t1 > 0.5) & (u1 < p)
        depth = -torch.log(torch.clamp(1.0 - u1, min=1e-12)) / kappa  # u1 < p  <=>  depth < L
        z = (t0 + depth) * dz[:, None]
        hit &= z > -h + 0.05
        t = torch.where(hit, t0 + depth, torch.full_like(L, INF))
        tb, jb = t.min(dim=1)
        take(tb, inten[jb], label)

    def porous_boxes(c, hx, hy, top, inten, label, kappa):
        c, hx, hy, top, inten = c.to(dev), hx.to(dev), hy.to(dev), top.to(dev), inten.to(dev)
        ox = c[:, 0][None, :] - sx
        oy = c[:, 1][None, :]
        idx_ = 1.0 / torch.where(dx.abs() < 1e-9, torch.full_like(dx, 1e-9), dx)[:, None]
        idy = 1.0 / torch.where(dy.abs() < 1e-9, torch.full_like(dy, 1e-9), dy)[:, None]
        idz = 1.0 / torch.where(dz.abs() < 1e-9, torch.full_like(dz, 1e-9), dz)[:, None]
        tx1, tx2 = (ox - hx[None, :]) * idx_, (ox + hx[None, :]) * idx_
        ty1, ty2 = (oy - hy[None, :]) * idy, (oy + hy[None, :]) * idy
        tz1, tz2 = ((-h + 0.05) * idz).expand_as(tx1), (-h + top)[None, :] * idz
        tmin = torch.maximum(torch.maximum(torch.minimum(tx1, tx2), torch.minimum(ty1, ty2)), torch.minimum(tz1, tz2))
        tmax = torch.minimum(torch.minimum(torch.maximum(tx1, tx2), torch.maximum(ty1, ty2)), torch.maximum(tz1, tz2))
        t0 = torch.clamp(tmin, min=0.5)
        L = torch.clamp(tmax - t0, min=0)
        u1 = torch.rand(L.shape, generator=g, device=dev)
        hit = (L > 0) & (u1 < 1.0 - torch.exp(-kappa * L))
        depth = -torch.log(torch.clamp(1.0 - u1, min=1e-12)) / kappa
        t = torch.where(hit, t0 + depth, torch.full_like(L, INF))
        tb, jb = t.min(dim=1)
        take(tb, inten[jb], label)

    seg0 = int(math.floor(sx / 100.0))
    for sg in (seg0 - 1, seg0, seg0 + 1):
        o = _segment_objects(seq, sg, kind)
        boxes(o["cars_c"], o["car_dx"], o["car_dy"], -h, -h + 1.5, o["car_i"], 10)
        boxes(o["walls_c"], o["wall_hx"], o["wall_hy"], -h, -h + o["wall_top"], o["wall_i"], 50)
        cylinders(o["cyl_c"], o["cyl_r"], o["cyl_h"], o["cyl_i"], 71)
        if "fence_c" in o:
            n_f = o["fence_c"].shape[0]
            boxes(o["fence_c"], o["fence_hx"], torch.full((n_f,), 0.05), -h, -h + 1.3, o["fence_i"], 51)
            boxes(o["clutter_c"], o["clutter_h"], o["clutter_h"], -h, -h + o["clutter_top"], o["clutter_i"], 99)
            porous(o["can_c"], o["can_r"], o["can_i"], 70, 0.9)
            porous(o["bush_c"], o["bush_r"], o["bush_i"], 70, 1.6)
            porous_boxes(o["hedge_c"], o["hedge_hx"], o["hedge_hy"], o["hedge_top"], o["hedge_i"], 70, 0.7)
        if "deck_c" in o:
            boxes(o["deck_c"], torch.full((1,), 44.0), torch.full((1,), 27.0), -h + 7.0, -h + 7.6, o["deck_i"], 52)
    mc, mi = _movers(seq, idx, kind)
    boxes(mc, torch.full((mc.shape[0],), 2.1), torch.full((mc.shape[0],), 0.9), -h, -h + 1.5, mi, 252)

    valid = best_t < 80.0
    if K["drop"] > 0:
        valid &= torch.rand(R, generator=g, device=dev) >= K["drop"]
    t = best_t + 0.02 * torch.randn(R, generator=g, device=dev)
    inten = torch.clamp(best_i + 2.0 * torch.randn(R, generator=g, device=dev), 0.0, 255.0)
    pts = torch.stack([t * dxs, t * dys, t * dzs, inten], 1)[valid].contiguous().float()
    labels = best_l[valid].contiguous() if with_labels else None
    return pts, labels, pose


def make_batch(seq, first, count, kind="K64", device="cpu", stride=1):
    """Concatenated scans + int32 offsets (host list) + poses."""
    pts, offs, poses, labels = [], [0], [], []
    for k in range(count):
        p, l, pose = make_scan(seq, first + k * stride, kind, device)
        pts.append(p)
        labels.append(l)
        offs.append(offs[-1] + p.shape[0])
        poses.append(pose)
    return torch.cat(pts, 0).contiguous(), offs, poses, torch.cat(labels, 0)
