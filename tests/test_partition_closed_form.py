"""The scene BVH's partition pass (reference Objects/BVH.cs:394-410) is a sequential two-pointer loop whose RESULT - the order
the items are left in - feeds the children's splits and the leaf order.  The device-side builder (csrc/ycge_bvh_build.hip) places
every item with a closed form instead of walking the loop; this file pins that closed form against the loop itself, exhaustively
for every class pattern up to 14 items and on random patterns up to 300."""
import itertools
import random


def two_pointer(is_left):
    """The reference's loop on item indices 0..n-1: returns (order, mid)."""
    a = list(range(len(is_left)))
    i0, i1 = 0, len(a) - 1
    while i0 <= i1:
        if is_left[a[i0]]:
            i0 += 1
        else:
            a[i0], a[i1] = a[i1], a[i0]
            i1 -= 1
    return a, i0


def closed_form(is_left):
    """What bvh_split_node computes, lane-parallel there: one prefix count of L's and the list of back L's."""
    n = len(is_left)
    e, n_left = n - 1, sum(is_left)
    mid = n_left
    if n_left == n:
        return list(range(n)), mid
    front_hi = mid if is_left[mid] else mid + 1              # front region [0, front_hi), back region [front_hi, n)
    pref_l = [0] * (n + 1)
    for p in range(n):
        pref_l[p + 1] = pref_l[p] + (1 if is_left[p] else 0)
    back_l = [None] * n
    for p in range(front_hi, n):
        if is_left[p]:
            back_l[n_left - pref_l[p] - 1] = p                # j - 1 = L's in (p, e]
    out = [None] * n
    for p in range(n):                                        # every p independently: this is the parallel loop
        if p < front_hi:
            if is_left[p]:
                out[p] = p
            else:
                k1 = p - pref_l[p]                            # R's in [0, p)
                dest = e if k1 == 0 else back_l[k1 - 1] - 1
                assert out[dest] is None
                out[dest] = p
                if p < mid:
                    assert out[p] is None
                    out[p] = back_l[k1]
        elif not is_left[p]:
            assert out[p - 1] is None
            out[p - 1] = p
    return out, mid


def test_closed_form_equals_the_loop_exhaustively():
    for n in range(1, 15):
        for cls in itertools.product((False, True), repeat=n):
            assert closed_form(cls) == two_pointer(cls), cls


def test_closed_form_equals_the_loop_on_random_patterns():
    rng = random.Random(20260210)
    for _ in range(4000):
        n = rng.randint(1, 300)
        pr = rng.random()
        cls = [rng.random() < pr for _ in range(n)]
        assert closed_form(cls) == two_pointer(cls)
