"""CPU restatement of the property the trace kernel's ray SEGMENTATION rests on (rt_wavefront.hip: plan_ray, segments_of, cut_at, axis_state_at):

The reference's grid walk (raytrace_opencl.c:383-398) is a 3-way merge of per-axis plane-crossing parameters
T_a(i) = (plane_a[i] - o_a) / d_a, evaluated in float32 exactly as the reference does.  Per axis the sequence is
non-decreasing, and every step takes the smallest head (x only if strictly smallest, else y if smaller than z, else z).
Hence the state of the walk once every crossing with T <= tau has been made is, per axis, just the NUMBER of such
crossings -- computable without walking.  This test walks step by step and compares, for many tau, the walked state with
the counted one, in numpy float32 (IEEE correctly rounded, like the kernel's divide)."""
import numpy as np
import pytest

F = np.float32
DIV = 256


def make_planes(rng):
    """257 increasing split planes per axis, unevenly spaced like vertex quantiles (trianglelist.cpp:657-678)."""
    p = np.sort(rng.random((3, DIV + 1)).astype(F) * F(4.0) - F(2.0), axis=1)
    return p


def crossing(planes, axis, o, d, cell_coord_after):
    """Head of `axis` when the walk stands in a cell with coordinate c on that axis: plane c+1 going up, c going down."""
    c = cell_coord_after
    idx = c + 1 if d[axis] >= 0 else c
    return F(F(planes[axis, idx] - o[axis]) / d[axis])


def walk(planes, o, d, start):
    """The reference's stepping; yields (cell tuple, heads) BEFORE each step, until the walk leaves the grid."""
    cell = list(start)
    heads = [crossing(planes, a, o, d, cell[a]) for a in range(3)]
    while True:
        yield tuple(cell), tuple(heads)
        dx, dy, dz = heads
        if dx < dy and dx < dz:
            a = 0
        elif dy < dz:
            a = 1
        else:
            a = 2
        cell[a] += 1 if d[a] >= 0 else -1
        if cell[a] < 0 or cell[a] >= DIV:
            return
        heads[a] = crossing(planes, a, o, d, cell[a])


def state_at(planes, o, d, start, tau):
    """Counted state: per axis, the number of crossings with T <= tau that stay inside the grid."""
    cell, heads = [], []
    for a in range(3):
        c0 = start[a]
        up = d[a] >= 0
        limit = (DIV - 1 - c0) if up else c0
        m = 0
        while m < limit:
            plane = c0 + m + 1 if up else c0 - m
            if F(F(planes[a, plane] - o[a]) / d[a]) <= tau:
                m += 1
            else:
                break
        c = c0 + m if up else c0 - m
        cell.append(c)
        heads.append(crossing(planes, a, o, d, c))
    return tuple(cell), tuple(heads)


@pytest.mark.parametrize("seed", range(6))
def test_counted_state_equals_walked_state(seed):
    rng = np.random.default_rng(seed)
    planes = make_planes(rng)
    checked = 0
    for _ in range(40):
        start = tuple(int(v) for v in rng.integers(0, DIV, 3))
        # origin inside the start cell, direction with every component non-zero (the only rays that are cut)
        o = np.array([planes[a, start[a]] + (planes[a, start[a] + 1] - planes[a, start[a]]) * F(rng.random()) for a in range(3)], dtype=F)
        d = (rng.random(3).astype(F) - F(0.5))
        d[np.abs(d) < F(1e-3)] = F(0.25)
        if seed % 2:  # some axis-dominant rays: long runs on one axis, many equal-looking heads
            d[rng.integers(0, 3)] *= F(50.0)
        steps = list(walk(planes, o, d, start))
        # exit parameter as plan_ray computes it
        te = min(F(F(planes[a, DIV if d[a] >= 0 else 0] - o[a]) / d[a]) for a in range(3))
        ta = min(steps[0][1])
        if not (np.isfinite(te) and np.isfinite(ta) and ta < te):
            continue
        for k in range(1, 9):
            tau = F(ta + F(te - ta) * F(k / 9.0))
            if not (ta <= tau < te):
                continue
            want = None
            for cell, heads in steps:  # the walk has made all crossings with T <= tau when its smallest head exceeds tau
                if min(heads) > tau:
                    want = (cell, heads)
                    break
            assert want is not None, "the walk left the grid before tau < te: the exit bound is wrong"
            got = state_at(planes, o, d, start, tau)
            assert got[0] == want[0], f"cell differs at tau={tau}: counted {got[0]} walked {want[0]}"
            assert all(np.float32(g) == np.float32(w) for g, w in zip(got[1], want[1])), f"heads differ at tau={tau}"
            checked += 1
    assert checked > 100


def test_ties_between_axes_are_all_before_the_cut():
    """Equal heads on two axes: the reference takes y before x and z before both (:387-398); whatever the order, all
    crossings with T == tau belong to the part of the walk before the cut, so counting `<= tau` per axis is consistent."""
    planes = np.tile(np.linspace(-1.0, 1.0, DIV + 1, dtype=F), (3, 1))
    o = np.array([planes[0, 10], planes[1, 20], planes[2, 30]], dtype=F)  # on plane corners: diagonal rays tie constantly
    d = np.array([1.0, 1.0, 1.0], dtype=F)
    start = (10, 20, 30)
    steps = list(walk(planes, o, d, start))
    for idx in range(5, len(steps) - 5, 7):
        tau = min(steps[idx][1])  # exactly a crossing value
        want = next((c, h) for c, h in steps if min(h) > tau)
        got = state_at(planes, o, d, start, tau)
        assert got[0] == want[0]
