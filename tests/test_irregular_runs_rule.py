"""The local rule of the generic clustering variant (csrc/scvod_k_cluster.inc, cc_run_is_plain), stated in Python and pinned
against the reference's loop (oracle_cluster = clusterAndCreateFrame, ssc.cpp:299-352) on the CPU.

"Everything found is joined" (connected components of the finds) can only differ from the loop where a find is one-sided:
around a run of points whose index triple lies outside the grid (or does not encode to its voxel key).  The rule looks at
the 3x3x3 cells around such a run's triple and around its key's own cell and says "settled" only when every one of those
finds is certain to stick; the test checks, per component of the finds, that a component whose irregular runs are all
settled is exactly one cluster of the reference.  (The device asks the same rule per run and falls into the exact
visiting-order rounds for the others; tests/test_gpu_parity.py::test_random_clouds_on_random_grids_cluster_fuzz holds the
device to the reference's partition.)"""
import numpy as np
import pytest


def _cloud(rng, cap, shrink):
    n = int(rng.integers(1, cap))
    kw = dict(range_res=float(rng.choice([0.05, 0.1, 0.2, 0.4, 0.8])), sector_res=float(rng.choice([0.3, 0.6, 1.2, 2.4])),
              azimuth_res=float(rng.choice([0.5, 1.0, 2.0, 4.0])))
    kind = rng.random(n)
    r = rng.uniform(0.5, 40, n)
    th = rng.uniform(0, 2 * np.pi, n)
    x = np.stack([r * np.cos(th), r * np.sin(th), rng.uniform(-3, 12, n), rng.uniform(0, 255, n)], 1)
    wall = kind < 0.5
    x[wall, 0] = np.round(x[wall, 0] / 6) * 6 + rng.normal(0, 0.05, wall.sum())
    x[kind > 0.98, 1] = 0.0   # polar angle exactly 0: sector index -1
    x[:, :3] *= shrink
    return kw, x.astype(np.float32)


def _canonical(labels):
    labels = np.asarray(labels)
    first = {}
    return np.array([first.setdefault(int(l), i) for i, l in enumerate(labels)])


def _groups(cs):
    """cells that hang together (adjacent cells list each other): cell -> group label"""
    lab = {c: j for j, c in enumerate(cs)}
    changed = True
    while changed:
        changed = False
        for a in cs:
            for b in cs:
                if max(abs(a[0] - b[0]), abs(a[1] - b[1]), abs(a[2] - b[2])) <= 1 and lab[b] < lab[a]:
                    lab[a] = lab[b]
                    changed = True
    return lab


def found_is_joined_and_rule(R, S, Az, apri):
    """-> (component of every point under "everything found is joined", roots of the components that hold a run the rule
    does not settle, runs looked at, runs settled)"""
    n = len(apri)
    ri, si, ai, key = (apri[k].astype(np.int64) for k in ("range_idx", "sector_idx", "azimuth_idx", "voxel_idx"))
    K = lambda c: c[0] * S + c[1] + c[2] * R * S
    reg = (ri >= 0) & (ri < R) & (si >= 0) & (si < S) & (ai >= 0) & (ai < Az) & (key == ri * S + si + ai * R * S)
    vox = {}
    for i in range(n):
        vox.setdefault(int(key[i]), []).append(i)
    regs = {k: [q for q in v if reg[q]] for k, v in vox.items()}

    def cells(r, s, a):  # findVoxelNeighbors (ssc.cpp:395-411): list order = range outermost, azimuth innermost
        return [(x, y, z) for x in range(r - 1, r + 2) if 0 <= x < R for y in range(s - 1, s + 2) if 0 <= y < S
                for z in range(a - 1, a + 2) if 0 <= z < Az]

    par = list(range(n))

    def find(a):
        while par[a] != a:
            par[a] = par[par[a]]
            a = par[a]
        return a

    def union(a, b):
        a, b = find(a), find(b)
        if a != b:
            par[max(a, b)] = min(a, b)

    found = set()
    for i in range(n):
        for c in cells(int(ri[i]), int(si[i]), int(ai[i])):
            if K(c) in vox:
                union(i, vox[K(c)][0])
                found.add(K(c))
    for k in found:  # every point of a voxel somebody lists is merged with it (ssc.cpp:316-345)
        for p in vox[k][1:]:
            union(p, vox[k][0])
    comp = np.array([find(i) for i in range(n)])

    irregular = np.nonzero(~reg)[0]
    unsettled, looked, settled, seen = set(), 0, 0, set()
    for i in irregular:
        t, hk = (int(ri[i]), int(si[i]), int(ai[i])), int(key[i])
        if (t, hk) in seen:  # (the device asks per run; the first point of the first such run is the strictest)
            continue
        seen.add((t, hk))
        looked += 1
        ok = True
        # (1) the listed voxels
        occ = [c for c in cells(*t) if K(c) in vox]
        if any(not regs[K(c)] for c in occ):
            ok = False
        elif occ:
            lab = _groups(occ)

            def untouched(c):
                if any(K(d) in vox and vox[K(d)][0] < i for d in cells(*c)):
                    return False
                return not any(q < i and c in cells(int(ri[q]), int(si[q]), int(ai[q])) for q in irregular)

            def labelled(c):
                if vox[K(c)][0] < i:
                    return True
                for d in cells(*c):
                    rp = regs.get(K(d), []) if d != c else []
                    if (len(rp) >= 3 and rp[2] < i) or (len(rp) >= 2 and rp[1] < i and c > d):
                        return True
                return False

            un = [untouched(c) for c in occ]
            sure = {lab[c] for j, c in enumerate(occ) if labelled(c) or all(un[j + 1:])}
            if any(lab[c] not in sure for c in occ):
                ok = False
        # (2) the home voxel
        hr = regs[hk]
        if not (len(hr) >= 2 or (len(hr) == 1 and i < hr[0]) or hk < 0 or hk >= R * S * Az):
            z, rem = divmod(hk, R * S)
            x, y = divmod(rem, S)
            around = [c for c in cells(x, y, z) if K(c) in vox and c != (x, y, z)]
            listers = [c for c in around if regs[K(c)]]
            strong = [c for c in listers if len(regs[K(c)]) >= 3 or (len(regs[K(c)]) >= 2 and (x, y, z) > c) or regs[K(c)][-1] > vox[hk][0]]
            if hr:
                if around and not strong:
                    ok = False
            elif listers:
                lab = _groups(listers)
                good = {lab[c] for c in strong}
                if any(lab[c] not in good for c in listers):
                    ok = False
        if ok:
            settled += 1
        else:
            unsettled.add(int(comp[i]))
    return comp, unsettled, looked, settled


@pytest.mark.parametrize("seed,shrink", [(77, 1.0), (5, 1.0), (21, 0.3), (23, 0.15)])
def test_components_the_rule_settles_are_clusters_of_the_reference(oracle, seed, shrink):
    from scvod_py import make_params as scvod_params
    rng = np.random.default_rng(seed)
    looked = settled = checked = level2 = 0
    for case in range(30):
        kw, x = _cloud(rng, 2500, shrink)
        P = scvod_params("semantickitti", **kw)
        apri = oracle.bin(P, x, case % 3 != 0)["apri"]
        if len(apri) == 0:
            continue
        R, S, Az = oracle.grid_dims(P)[:3]
        comp, unsettled, a, b = found_is_joined_and_rule(R, S, Az, apri)
        looked += a
        settled += b
        ref = _canonical(oracle.cluster(P, apri)[0])
        fj = _canonical(comp)
        # the reference never joins what nobody found
        pairs = np.unique(np.stack([ref, fj], 1), axis=0)
        assert len(np.unique(pairs[:, 0])) == len(pairs), f"case {case}: a reference cluster spans two components of the finds"
        ok = ~np.isin(comp, list(unsettled))
        checked += int(ok.sum())
        assert np.array_equal(fj[ok], ref[ok]), f"seed {seed} case {case} {kw}: the rule settled a run whose finds the loop drops"
    assert looked > 50 and settled > looked // 2 and checked > 0


def test_the_cases_the_rule_must_not_settle(oracle):
    """hand-built: (a) a passenger behind the only regular point of its home voxel, whose lister adopts a label further
    down its list (the passenger is left alone for good: the rule must say "not settled"); (b) the same passenger in front
    of the regular point (visited first: settled)."""
    from scvod_py import APRI_DTYPE, make_params
    P = make_params("semantickitti")
    R, S, Az = oracle.grid_dims(P)[:3]

    def apri_of(triples_keys):
        a = np.zeros(len(triples_keys), APRI_DTYPE)
        for i, (t, k) in enumerate(triples_keys):
            a["range_idx"][i], a["sector_idx"][i], a["azimuth_idx"][i] = t
            a["voxel_idx"][i] = k
        return a

    K = lambda c: c[0] * S + c[1] + c[2] * R * S
    home, far = (29, 268, 20), (30, 268, 21)
    out = (R + 30, 268, 19)  # range index beyond the grid: lists nothing
    lone = (R + 31, 268, 20)
    # a pure-irregular voxel on `far`'s key (visited first, alone), the regular point of `home`, then its passenger
    pts = [(lone, K(far)), (home, K(home)), (out, K(home))]
    apri = apri_of(pts)
    comp, unsettled, looked, settled = found_is_joined_and_rule(R, S, Az, apri)
    ref = _canonical(oracle.cluster(P, apri)[0])
    assert len(set(comp)) == 1 and len(set(ref)) == 2 and ref[2] == 2  # found-is-joined has one cluster, the loop leaves the passenger out
    assert int(comp[2]) in unsettled
    apri = apri_of([pts[0], pts[2], pts[1]])
    comp, unsettled, looked, settled = found_is_joined_and_rule(R, S, Az, apri)
    ref = _canonical(oracle.cluster(P, apri)[0])
    assert len(set(ref)) == 1 and len(set(comp)) == 1
    assert int(comp[1]) not in unsettled  # (visited before the regular point: its visit finds it labelled)
