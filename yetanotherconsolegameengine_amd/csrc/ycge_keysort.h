// ycge_keysort.h - .NET 8 Array.Sort over a sub-range, as an index sort; shared by the host builders (ycge_accel.cpp)
// and the device-side scene-BVH builder (ycge_bvh_build.hip), where one lane runs it.
//
// The reference falls back to Array.Sort(arr, start, count, comparer) on the centroid of one axis when binning
// finds no split or the partition leaves a side empty (BVH.cs:389,419; MeshBVH.cs:506,536).  Array.Sort is
// System.Private.CoreLib's introspective sort (ArraySortHelper<T>): depth limit 2*(log2(n)+1); partitions of
// <= 16 are finished by insertion sort (2 and 3 by compare-exchange), pivot = median of first/middle/last parked
// at hi-1, heapsort when the depth budget is spent.  It is not stable, so the order it leaves equal keys in is
// reproduced by running the same procedure.  IntroSort's recursion (right part first, then loop on the left) is
// an explicit stack here: the two parts are disjoint, so the order they are finished in does not change the result.
#pragma once
#include <stdint.h>

#include "ycge_math.h"

namespace ycge {

template <class IdxT> struct KeySorter {
    IdxT *ord;              // permutation slice being sorted
    const float *key;       // centroid of the chosen axis, indexed by item id

    YCGE_HD int cmp(IdxT a, IdxT b) const       // float.CompareTo
    {
        const float x = key[a], y = key[b];
        if (x < y) return -1;
        if (x > y) return 1;
        if (x == y) return 0;
        if (is_nan(x)) return is_nan(y) ? 0 : -1;
        return 1;
    }
    YCGE_HD void exch(int i, int j) { const IdxT t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    YCGE_HD void order2(int lo, int i, int j) { if (cmp(ord[lo + i], ord[lo + j]) > 0) exch(lo + i, lo + j); }

    YCGE_HD void insertion(int lo, int n)
    {
        for (int i = 0; i + 1 < n; i++) {
            const IdxT t = ord[lo + i + 1];
            int j = i;
            for (; j >= 0 && cmp(t, ord[lo + j]) < 0; j--) ord[lo + j + 1] = ord[lo + j];
            ord[lo + j + 1] = t;
        }
    }
    YCGE_HD void sift(int lo, int i, int n)     // 1-based heap positions
    {
        const IdxT d = ord[lo + i - 1];
        while (i <= n / 2) {
            int child = 2 * i;
            if (child < n && cmp(ord[lo + child - 1], ord[lo + child]) < 0) child++;
            if (!(cmp(d, ord[lo + child - 1]) < 0)) break;
            ord[lo + i - 1] = ord[lo + child - 1];
            i = child;
        }
        ord[lo + i - 1] = d;
    }
    YCGE_HD void heap(int lo, int n)
    {
        for (int i = n / 2; i >= 1; i--) sift(lo, i, n);
        for (int i = n; i > 1; i--) { exch(lo, lo + i - 1); sift(lo, 1, i - 1); }
    }
    YCGE_HD int partition(int lo, int n)
    {
        const int hi = n - 1, mid = hi >> 1;
        order2(lo, 0, mid);
        order2(lo, 0, hi);
        order2(lo, mid, hi);
        const IdxT pivot = ord[lo + mid];
        exch(lo + mid, lo + hi - 1);
        int left = 0, right = hi - 1;
        while (left < right) {
            while (cmp(ord[lo + (++left)], pivot) < 0) {}
            while (cmp(pivot, ord[lo + (--right)]) < 0) {}
            if (left >= right) break;
            exch(lo + left, lo + right);
        }
        if (left != hi - 1) exch(lo + left, lo + hi - 1);
        return left;
    }
    YCGE_HD void sort(int lo0, int n0)
    {
        if (n0 < 2) return;
        int lg = 0;
        for (uint32_t v = (uint32_t)n0; v >>= 1;) lg++;
        // pending parts: at most one per level of the depth budget (<= 2 * 32 + 2)
        int st_lo[72], st_n[72], st_depth[72], sp = 0;
        st_lo[0] = lo0; st_n[0] = n0; st_depth[0] = 2 * (lg + 1); sp = 1;
        while (sp > 0) {
            sp--;
            int lo = st_lo[sp], n = st_n[sp], depth = st_depth[sp];
            while (n > 1) {
                if (n <= 16) {
                    if (n == 2) order2(lo, 0, 1);
                    else if (n == 3) { order2(lo, 0, 1); order2(lo, 0, 2); order2(lo, 1, 2); }
                    else insertion(lo, n);
                    break;
                }
                if (depth == 0) { heap(lo, n); break; }
                depth--;
                const int p = partition(lo, n);
                st_lo[sp] = lo + p + 1; st_n[sp] = n - (p + 1); st_depth[sp] = depth; sp++;      // the right part, later
                n = p;
            }
        }
    }
};

} // namespace ycge
