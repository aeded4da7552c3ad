// select_replay.h -- device-side replay of KeyPointsFilter::retainBest (OpenCV features2d keypoint.cpp),
// i.e. std::nth_element + std::partition, reproducing the exact permutation of the C++ standard library
// the cv2 wheel was linked against.  The SET retainBest keeps is order-independent; the ORDER it leaves
// the survivors in (= the order of cv2's keypoints, hence every queryIdx/trainIdx downstream) is the
// standard library's introselect permutation, so it is restated here step by step:
//   * MO_ORDER_LIBSTDCXX: libstdc++ __introselect (median-of-3 to first, unguarded Hoare partition,
//     heap-select fallback at depth 2*lg(n), insertion sort of the last <= 3)
//   * MO_ORDER_MSVC: MSVC STL nth_element (ninther / median-of-3 guess, three-way partition around the
//     pivot's equal range, insertion sort of the last <= 32) -- the order of the reference's gt.yaml.
// Reference call sites: src/orbslam2/extractor.py:65 (orb.detectAndCompute -> computeKeyPoints ->
// retainBest twice per level).
//
// These are sequential algorithms; one lane replays them on a record array that lives in LDS (or in an
// HBM scratch slot when the level has more candidates than the LDS window).  Records:
//   uint32_t: FAST pass   -- score in bits 31..24, packed (y<<12|x) position in bits 23..0
//   uint64_t: Harris pass -- float response bits in 63..32, packed position in 31..0
#pragma once
#include <stdint.h>

namespace replay {

template <class T> struct Rec;
template <> struct Rec<uint32_t> {
    static __device__ __forceinline__ bool gt(uint32_t a, uint32_t b) { return (a >> 24) > (b >> 24); }
    static __device__ __forceinline__ bool ge(uint32_t a, uint32_t b) { return (a >> 24) >= (b >> 24); }
};
template <> struct Rec<uint64_t> {
    static __device__ __forceinline__ float f(uint64_t a) { return __uint_as_float((uint32_t)(a >> 32)); }
    static __device__ __forceinline__ bool gt(uint64_t a, uint64_t b) { return f(a) > f(b); }
    static __device__ __forceinline__ bool ge(uint64_t a, uint64_t b) { return f(a) >= f(b); }
};

template <class P> __device__ __forceinline__ void swp(P a, int i, int j) {
    auto t = a[i];
    a[i] = a[j];
    a[j] = t;
}

// ---------------------------------------------------------------- libstdc++ ------------------------
template <class T, class P> __device__ __forceinline__ void ls_move_median_to_first(P a, int result, int ia, int ib, int ic) {
    typedef Rec<T> R;
    if (R::gt(a[ia], a[ib])) {
        if (R::gt(a[ib], a[ic])) swp(a, result, ib);
        else if (R::gt(a[ia], a[ic])) swp(a, result, ic);
        else swp(a, result, ia);
    } else if (R::gt(a[ia], a[ic])) swp(a, result, ia);
    else if (R::gt(a[ib], a[ic])) swp(a, result, ic);
    else swp(a, result, ib);
}

template <class T, class P> __device__ __forceinline__ int ls_unguarded_partition(P a, int first, int last, int pivot) {
    typedef Rec<T> R;
    const T pv = a[pivot];
    while (true) {
        while (R::gt(a[first], pv)) ++first;
        --last;
        while (R::gt(pv, a[last])) --last;
        if (!(first < last)) return first;
        swp(a, first, last);
        ++first;
    }
}

template <class T, class P> __device__ __forceinline__ void ls_push_heap(P a, int first, int hole, int top, T value) {
    typedef Rec<T> R;
    int parent = (hole - 1) / 2;
    while (hole > top && R::gt(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

template <class T, class P> __device__ __forceinline__ void ls_adjust_heap(P a, int first, int hole, int len, T value) {
    typedef Rec<T> R;
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (R::gt(a[first + second], a[first + (second - 1)])) second--;
        a[first + hole] = a[first + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        a[first + hole] = a[first + (second - 1)];
        hole = second - 1;
    }
    ls_push_heap<T>(a, first, hole, top, value);
}

template <class T, class P> __device__ __forceinline__ void ls_heap_select(P a, int first, int middle, int last) {
    typedef Rec<T> R;
    // __make_heap(first, middle)
    int len = middle - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            T v = a[first + parent];
            ls_adjust_heap<T>(a, first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int i = middle; i < last; ++i) {
        if (R::gt(a[i], a[first])) {
            // __pop_heap(first, middle, i)
            T v = a[i];
            a[i] = a[first];
            ls_adjust_heap<T>(a, first, 0, middle - first, v);
        }
    }
}

template <class T, class P> __device__ __forceinline__ void ls_insertion_sort(P a, int first, int last) {
    typedef Rec<T> R;
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        T val = a[i];
        if (R::gt(val, a[first])) {
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else {
            int l = i, next = i - 1;
            while (R::gt(val, a[next])) {
                a[l] = a[next];
                l = next;
                --next;
            }
            a[l] = val;
        }
    }
}

template <class T, class P> __device__ __forceinline__ void ls_introselect(P a, int first, int nth, int last, int depth) {
    while (last - first > 3) {
        if (depth == 0) {
            ls_heap_select<T>(a, first, nth + 1, last);
            swp(a, first, nth);
            return;
        }
        --depth;
        int mid = first + (last - first) / 2;
        ls_move_median_to_first<T>(a, first, first + 1, mid, last - 1);
        int cut = ls_unguarded_partition<T>(a, first + 1, last, first);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    ls_insertion_sort<T>(a, first, last);
}

template <class T, class P> __device__ __forceinline__ void ls_nth_element(P a, int first, int nth, int last) {
    if (first == last || nth == last) return;
    ls_introselect<T>(a, first, nth, last, (31 - __clz(last - first)) * 2);
}

// ---------------------------------------------------------------- MSVC STL -------------------------
template <class T, class P> __device__ __forceinline__ void ms_med3(P a, int first, int mid, int last) {
    typedef Rec<T> R;
    if (R::gt(a[mid], a[first])) swp(a, mid, first);
    if (R::gt(a[last], a[mid])) {
        swp(a, last, mid);
        if (R::gt(a[mid], a[first])) swp(a, mid, first);
    }
}

template <class T, class P> __device__ __forceinline__ void ms_guess_median(P a, int first, int mid, int last) {  // last inclusive
    const int count = last - first;
    if (40 < count) {
        const int step = (count + 1) >> 3;
        const int two_step = step << 1;
        ms_med3<T>(a, first, first + step, first + two_step);
        ms_med3<T>(a, mid - step, mid, mid + step);
        ms_med3<T>(a, last - two_step, last - step, last);
        ms_med3<T>(a, first + step, mid, last - step);
    } else {
        ms_med3<T>(a, first, mid, last);
    }
}

template <class T, class P> __device__ __forceinline__ void ms_partition(P a, int first, int last, int& out_first, int& out_last) {
    typedef Rec<T> R;
    int mid = first + ((last - first) >> 1);
    ms_guess_median<T>(a, first, mid, last - 1);
    int pfirst = mid;
    int plast = pfirst + 1;
    while (first < pfirst && !R::gt(a[pfirst - 1], a[pfirst]) && !R::gt(a[pfirst], a[pfirst - 1])) --pfirst;
    while (plast < last && !R::gt(a[plast], a[pfirst]) && !R::gt(a[pfirst], a[plast])) ++plast;
    int gfirst = plast;
    int glast = pfirst;
    for (;;) {
        for (; gfirst < last; ++gfirst) {
            if (R::gt(a[pfirst], a[gfirst])) continue;
            else if (R::gt(a[gfirst], a[pfirst])) break;
            else if (plast != gfirst) { swp(a, plast, gfirst); ++plast; }
            else ++plast;
        }
        for (; first < glast; --glast) {
            if (R::gt(a[glast - 1], a[pfirst])) continue;
            else if (R::gt(a[pfirst], a[glast - 1])) break;
            else if (--pfirst != glast - 1) swp(a, pfirst, glast - 1);
        }
        if (glast == first && gfirst == last) { out_first = pfirst; out_last = plast; return; }
        if (glast == first) {
            if (plast != gfirst) swp(a, pfirst, plast);
            ++plast;
            swp(a, pfirst, gfirst);
            ++pfirst;
            ++gfirst;
        } else if (gfirst == last) {
            if (--glast != --pfirst) swp(a, glast, pfirst);
            swp(a, pfirst, --plast);
        } else {
            swp(a, gfirst, --glast);
            ++gfirst;
        }
    }
}

template <class T, class P> __device__ __forceinline__ void ms_insertion_sort(P a, int first, int last) {
    typedef Rec<T> R;
    if (first == last) return;
    for (int mid = first + 1; mid != last; ++mid) {
        int hole = mid;
        T val = a[mid];
        if (R::gt(val, a[first])) {
            for (int k = mid; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else {
            int prev = hole - 1;
            while (R::gt(val, a[prev])) {
                a[hole] = a[prev];
                hole = prev;
                --prev;
            }
            a[hole] = val;
        }
    }
}

template <class T, class P> __device__ __forceinline__ void ms_nth_element(P a, int first, int nth, int last) {
    if (nth == last) return;
    while (32 < last - first) {
        int mf, ml;
        ms_partition<T>(a, first, last, mf, ml);
        if (ml <= nth) first = ml;
        else if (mf <= nth) return;
        else last = mf;
    }
    ms_insertion_sort<T>(a, first, last);
}

// std::partition (bidirectional form, same permutation in libstdc++ and the MSVC STL):
// elements with response >= thr first.  Returns the number of elements satisfying the predicate.
template <class T, class P> __device__ __forceinline__ int partition_ge(P a, int first, int last, T thr) {
    typedef Rec<T> R;
    const int begin = first;
    while (true) {
        while (true) {
            if (first == last) return first - begin;
            else if (R::ge(a[first], thr)) ++first;
            else break;
        }
        --last;
        while (true) {
            if (first == last) return first - begin;
            else if (!R::ge(a[last], thr)) --last;
            else break;
        }
        swp(a, first, last);
        ++first;
    }
}

// KeyPointsFilter::retainBest on records a[0..n): returns the surviving count, survivors in a[0..ret)
template <class T, class P> __device__ __forceinline__ int retain_best(P a, int n, int n_points, int order) {
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    if (order == MO_ORDER_MSVC) ms_nth_element<T>(a, 0, n_points - 1, n);
    else ls_nth_element<T>(a, 0, n_points - 1, n);
    T amb = a[n_points - 1];
    return n_points + partition_ge<T>(a, n_points, n, amb);
}


// ================================================================ wave-parallel replay (libstdc++ order) ============
// The Hoare partition of libstdc++'s introselect is a pairing: the k-th element from the left that stops the
// upward scan (not better than the pivot) is swapped with the k-th element from the right that stops the downward
// scan (not worse than the pivot), for as long as the left one lies before the right one.  Ranks are prefix counts,
// so one wavefront computes the whole partition in three passes of ballots + popcounts instead of a serial walk,
// and reproduces the serial permutation exactly:
//   L_k = k-th position from the left with !(a > pivot),  R_k = k-th from the right with !(pivot > a)
//   K   = #{k : L_k < R_k};  swap a[L_k] <-> a[R_k] for k <= K;  cut = min(L_{K+1}, R_K)
// All 64 lanes of ONE wavefront call these functions convergently.  rpos: u16[>= n/2], bl: u64[>= n/64 + 8].
// The replay runs on ONE wavefront of a (possibly larger) workgroup, so it must not use workgroup barriers.  Lanes of a
// wavefront execute in lockstep and LDS / same-CU global accesses of one wavefront are served in issue order; what is
// needed between a store by one lane and a load of it by another is that the compiler keeps the order and the stores
// have left the wavefront: a workgroup-scope fence (s_waitcnt, no s_barrier).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#define REPLAY_SERIAL_BELOW 32  // ranges this short are finished by one lane (a wave partition step costs ~1.5k cycles)

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// generic pairing partition: stopL(v) / stopR(v) classify an element; returns cut (absolute index) and total of R.
// Each lane handles 4 consecutive elements per trip (256 per wavefront trip): element index = c0 + 4*lane + k, so the
// rank of an element is (stoppers in earlier trips) + (stoppers of lower lanes, any k) + (own stoppers with smaller k).
// bl: one ballot per (trip, k) -> u64[4 * ceil(n / 256)].
#define RP_EPL 4
template <class T, class P, class FL, class FR>
__device__ __forceinline__ int wave_pair_partition(P a, int lo, int hi, FL stopL, FR stopR, uint16_t* rpos, unsigned long long* bl,
                                   int lane, int* total_r) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    int TR = 0;
    for (int c0 = lo, ch = 0; c0 < hi; c0 += 64 * RP_EPL, ch++) {
        T v[RP_EPL];
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            int i = c0 + RP_EPL * lane + k;
            v[k] = a[i < hi ? i : lo];
        }
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            bool in = c0 + RP_EPL * lane + k < hi;
            unsigned long long mL = __ballot(in && stopL(v[k])), mR = __ballot(in && stopR(v[k]));
            if (lane == 0) bl[ch * RP_EPL + k] = mL;
            TR += __popcll(mR);
        }
    }
    // cut = min(first non-swapping left stopper, last swapping right stopper = smallest swapping-R position): positions grow
    // with (trip, lane, k), so the first trip in which any lane holds one decides, and the lowest such lane holds the minimum
    int baseL = 0, baseR = 0, K = 0, minNL = 0x7FFFFFFF, minSR = 0x7FFFFFFF;
    bool foundNL = false, foundSR = false;
    for (int c0 = lo; c0 < hi; c0 += 64 * RP_EPL) {
        T v[RP_EPL];
        int nl_pos = 0x7FFFFFFF, sr_pos = 0x7FFFFFFF;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            int i = c0 + RP_EPL * lane + k;
            v[k] = a[i < hi ? i : lo];
        }
        bool isL[RP_EPL], isR[RP_EPL];
        int pl = 0, pr = 0, tl = 0, tr = 0;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            bool in = c0 + RP_EPL * lane + k < hi;
            isL[k] = in && stopL(v[k]);
            isR[k] = in && stopR(v[k]);
            unsigned long long mL = __ballot(isL[k]), mR = __ballot(isR[k]);
            pl += __popcll(mL & lt); pr += __popcll(mR & lt);
            tl += __popcll(mL); tr += __popcll(mR);
        }
        int cL = baseL + pl, cR = baseR + pr;   // stoppers strictly before this lane's first element
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            const int i = c0 + RP_EPL * lane + k;
            const int kL = cL + 1, kR = TR - cR;
            const bool swL = isL[k] && (TR - cR - (isR[k] ? 1 : 0)) >= kL;   // R_kL lies strictly right of this element
            const bool swR = isR[k] && cL >= kR;                              // L_kR lies strictly left of this element
            if (swR) rpos[kR - 1] = (uint16_t)(i - lo);
            if (isL[k] && !swL) nl_pos = min(nl_pos, i);
            if (swR) sr_pos = min(sr_pos, i);
            K += __popcll(__ballot(swL));
            cL += isL[k] ? 1 : 0;
            cR += isR[k] ? 1 : 0;
        }
        if (!foundNL) {
            unsigned long long mk = __ballot(nl_pos != 0x7FFFFFFF);
            if (mk) { minNL = __builtin_amdgcn_readlane(nl_pos, __ffsll((long long)mk) - 1); foundNL = true; }
        }
        if (!foundSR) {
            unsigned long long mk = __ballot(sr_pos != 0x7FFFFFFF);
            if (mk) { minSR = __builtin_amdgcn_readlane(sr_pos, __ffsll((long long)mk) - 1); foundSR = true; }
        }
        baseL += tl;
        baseR += tr;
    }
    wave_sync();
    baseL = 0;
    for (int c0 = lo, ch = 0; c0 < hi && baseL < K; c0 += 64 * RP_EPL, ch++) {
        unsigned long long mL[RP_EPL];
        int pl = 0, tl = 0;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            mL[k] = bl[ch * RP_EPL + k];
            pl += __popcll(mL[k] & lt);
            tl += __popcll(mL[k]);
        }
        int cL = baseL + pl;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            if ((mL[k] >> lane) & 1ull) {
                const int kL = cL + 1;
                if (kL <= K) {
                    int p = c0 + RP_EPL * lane + k, q = lo + rpos[kL - 1];
                    T vp = a[p], vq = a[q];
                    a[p] = vq;
                    a[q] = vp;
                }
                cL++;
            }
        }
        baseL += tl;
    }
    wave_sync();
    if (total_r) *total_r = TR;
    return min(minNL, minSR);
}

// ---------------------------------------------------------------- register-resident tail (libstdc++ order) ---------
// A range of at most 64 records fits one wavefront with ONE record per lane, and from there on the rest of introselect runs on
// registers: the median-of-3 reads are v_readlane, the Hoare pairing is two ballots and popcounts (rank of a stopper = stoppers of
// lower lanes), the swaps are one ds_permute (every swapping record pushes itself to a slot lane: left swapper of rank k to lane k,
// right swapper of rank k to lane 32 + k; a range of <= 63 partitioned records has at most 31 pairs) and one ds_bpermute (it pulls
// its partner's record from the opposite slot), the final insertion sort of <= 3 records is scalar code on broadcast values.  About
// 400 cycles per round against 1.5 - 2 k for a round through LDS arrays, and no serial lane-0 tail (ranges below 32 records were
// finished by ONE lane walking LDS: 5 - 10 k cycles); typically the last five or six rounds of every replay.
template <class T> struct LaneOps;
template <> struct LaneOps<uint32_t> {
    static __device__ __forceinline__ uint32_t read(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }
    static __device__ __forceinline__ uint32_t push(uint32_t v, int dst) { return (uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)v); }
    static __device__ __forceinline__ uint32_t pull(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); }
};
template <> struct LaneOps<uint64_t> {
    static __device__ __forceinline__ uint64_t read(uint64_t v, int src) {
        const int s = __builtin_amdgcn_readfirstlane(src);
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), s) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, s);
    }
    static __device__ __forceinline__ uint64_t push(uint64_t v, int dst) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)(uint32_t)v);
    }
    static __device__ __forceinline__ uint64_t pull(uint64_t v, int src) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)v);
    }
};

#define REPLAY_REG_MAX 64  // ranges this short (and longer than 3) finish in registers

// continuation of ls_introselect on [first, last), 3 < last - first <= 64; all 64 lanes of ONE wavefront, convergent
template <class T, class P>
__device__ __forceinline__ void wave_reg_introselect(P a, int first, int nth, int last, int depth, int lane) {
    typedef Rec<T> R;
    typedef LaneOps<T> L;
    const int n = last - first, kth = nth - first;
    T v = a[first + min(lane, n - 1)];
    int lo = 0, hi = n;
    const unsigned long long lt = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~0ull << (lane + 1);
    while (hi - lo > 3) {
        if (depth == 0) {  // heap-select fallback (never seen on image data): back to the array, one lane
            if (lane < n) a[first + lane] = v;
            wave_sync();
            if (lane == 0) { ls_heap_select<T>(a, first + lo, nth + 1, first + hi); swp(a, first + lo, nth); }
            wave_sync();
            return;
        }
        --depth;
        const int mid = lo + (hi - lo) / 2;
        // __move_median_to_first(result = lo, lo + 1, mid, hi - 1)
        const T va = L::read(v, lo + 1), vb = L::read(v, mid), vc = L::read(v, hi - 1);
        int p;
        if (R::gt(va, vb)) p = R::gt(vb, vc) ? mid : R::gt(va, vc) ? hi - 1 : lo + 1;
        else p = R::gt(va, vc) ? lo + 1 : R::gt(vb, vc) ? hi - 1 : mid;
        const T vlo = L::read(v, lo), pv = L::read(v, p);
        if (lane == lo) v = pv;
        else if (lane == p) v = vlo;
        // __unguarded_partition of [lo + 1, hi) around pv, as the rank pairing of wave_pair_partition
        const bool in = lane > lo && lane < hi;
        const bool isL = in && !R::gt(v, pv), isR = in && !R::gt(pv, v);
        const unsigned long long mL = __ballot(isL), mR = __ballot(isR);
        const int cL = __popcll(mL & lt), cRa = __popcll(mR & above);  // left stoppers strictly before / right stoppers strictly after this lane
        const bool swL = isL && cRa >= cL + 1, swR = isR && cL >= cRa + 1;
        const unsigned long long sL = __ballot(swL), sR = __ballot(swR);
        if (sL) {  // wave-uniform
            const T got = L::push(v, swL ? cL : swR ? 32 + cRa : 63);   // (lane 63 is no slot: at most 31 pairs)
            const T mine = L::pull(got, swL ? 32 + cL : swR ? cRa : lane);
            if (swL || swR) v = mine;
        }
        const unsigned long long nl = mL & ~sL;
        const int minNL = nl ? __ffsll((long long)nl) - 1 : 0x7FFFFFFF, minSR = sR ? __ffsll((long long)sR) - 1 : 0x7FFFFFFF;
        const int cut = min(minNL, minSR);
        if (cut <= kth) lo = cut;
        else hi = cut;
    }
    // __insertion_sort of the last <= 3 records
    const int m = hi - lo;
    if (m >= 2) {
        T e0 = L::read(v, lo), e1 = L::read(v, lo + 1), e2 = m == 3 ? L::read(v, lo + 2) : e1;
        if (R::gt(e1, e0)) { const T t = e0; e0 = e1; e1 = t; }
        if (m == 3) {
            if (R::gt(e2, e0)) { const T t = e2; e2 = e1; e1 = e0; e0 = t; }
            else if (R::gt(e2, e1)) { const T t = e1; e1 = e2; e2 = t; }
        }
        if (lane == lo) v = e0;
        else if (lane == lo + 1) v = e1;
        else if (m == 3 && lane == lo + 2) v = e2;
    }
    if (lane < n) a[first + lane] = v;
    wave_sync();
}

// continuation of libstdc++'s __introselect on [first, last) with `depth` rounds left before the heap-select fallback; all 64 lanes of ONE
// wavefront, convergent, no workgroup barrier
template <class T, class P>
__device__ __forceinline__ void wave_ls_introselect(P a, int first, int nth, int last, int depth, uint16_t* rpos, unsigned long long* bl, int lane) {
    typedef Rec<T> R;
    while (last - first > 3) {
        if (last - first <= REPLAY_REG_MAX) { wave_reg_introselect<T>(a, first, nth, last, depth, lane); return; }
        if (last - first > 65535 || depth == 0) {
            if (lane == 0) ls_introselect<T>(a, first, nth, last, depth);  // identical continuation, one lane
            wave_sync();
            return;
        }
        --depth;
        int mid = first + (last - first) / 2;
        if (lane == 0) ls_move_median_to_first<T>(a, first, first + 1, mid, last - 1);
        wave_sync();
        const T pv = a[first];
        int cut = wave_pair_partition<T>(
            a, first + 1, last, [pv](T v) { return !R::gt(v, pv); }, [pv](T v) { return !R::gt(pv, v); }, rpos, bl, lane,
            (int*)nullptr);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0) ls_insertion_sort<T>(a, first, last);
    wave_sync();
}

template <class T, class P>
__device__ __forceinline__ void wave_ls_nth_element(P a, int first, int nth, int last, uint16_t* rpos, unsigned long long* bl, int lane) {
    if (first == last || nth == last) return;
    wave_ls_introselect<T>(a, first, nth, last, (31 - __clz(last - first)) * 2, rpos, bl, lane);
}

// retainBest, all lanes convergent.  libstdc++ order runs wave-parallel; the MSVC STL's three-way partition is
// replayed by one lane.
template <class T, class P>
__device__ __forceinline__ int wave_retain_best(P a, int n, int n_points, int order, uint16_t* rpos, unsigned long long* bl, int lane) {
    typedef Rec<T> R;
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    if (order == MO_ORDER_MSVC) {
        if (lane == 0) ms_nth_element<T>(a, 0, n_points - 1, n);
        wave_sync();
    } else {
        wave_ls_nth_element<T>(a, 0, n_points - 1, n, rpos, bl, lane);
    }
    const T amb = a[n_points - 1];
    int tail = n - n_points;
    if (tail < REPLAY_SERIAL_BELOW || tail > 65535) {
        int keep = 0;
        if (lane == 0) keep = partition_ge<T>(a, n_points, n, amb);
        wave_sync();
        return n_points + __shfl(keep, 0, 64);
    }
    int total_true = 0;
    wave_pair_partition<T>(
        a, n_points, n, [amb](T v) { return !R::ge(v, amb); }, [amb](T v) { return R::ge(v, amb); }, rpos, bl, lane,
        &total_true);
    return n_points + total_true;
}

// ================================================================ workgroup-parallel replay (4 wavefronts) ==========
// The same pairing partition with the 256-element trips dealt round-robin to the NT / 64 wavefronts of the workgroup.
// Trip t (elements lo + 256 t ..) needs the number of left / right stoppers in the trips before it: pass 1 leaves the
// per-trip counts in LDS, one wavefront turns them into exclusive prefix sums (<= 64 trips: one shuffle scan), passes 2
// and 3 then run independently per trip.  K and the cut are combined over the wavefronts (positions grow with the trip
// number, so the minimum over the wavefronts' first hits is the global first hit).  All 256 threads call these functions
// convergently; every branch below depends only on values that are identical in all threads.
// NT = threads of the workgroup (k_select runs 256 in the batched mode - 3 workgroups per CU - and 1024 for one or two frames, where
// nothing else competes for the CU and the long first partition rounds of a level are what the call waits for)
template <int NT> struct WgScratch {
    int cl[64], cr[64], pl[65], pr[65];
    int k[NT / 64], nl[NT / 64], sr[NT / 64];
    int cut;
};
#define WG_PARTITION_MIN 384     // shorter ranges: one wavefront does it alone (saves the barriers)
#define WG_PARTITION_MAX 16384   // 64 trips: what one shuffle scan covers

template <int NT, class T, class P, class FL, class FR>
__device__ __forceinline__ int wg_pair_partition(P a, int lo, int hi, FL stopL, FR stopR, uint16_t* rpos, unsigned long long* bl,
                                                 int tid, WgScratch<NT>* ws, int* total_r) {
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int ntrip = (hi - lo + 64 * RP_EPL - 1) / (64 * RP_EPL);
    // pass 1: ballots of the left stoppers (kept for pass 3) and per-trip stopper counts
    for (int t = wv; t < ntrip; t += NT / 64) {
        const int c0 = lo + t * 64 * RP_EPL;
        T v[RP_EPL];
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            int i = c0 + RP_EPL * lane + k;
            v[k] = a[i < hi ? i : lo];
        }
        int tl = 0, tr = 0;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            bool in = c0 + RP_EPL * lane + k < hi;
            unsigned long long mL = __ballot(in && stopL(v[k])), mR = __ballot(in && stopR(v[k]));
            if (lane == 0) bl[t * RP_EPL + k] = mL;
            tl += __popcll(mL);
            tr += __popcll(mR);
        }
        if (lane == 0) { ws->cl[t] = tl; ws->cr[t] = tr; }
    }
    __syncthreads();
    if (wv == 0) {  // exclusive prefix sums over the trips
        int l = lane < ntrip ? ws->cl[lane] : 0, r = lane < ntrip ? ws->cr[lane] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int nl = __shfl_up(l, o, 64), nr = __shfl_up(r, o, 64);
            if (lane >= o) { l += nl; r += nr; }
        }
        ws->pl[lane + 1] = l;
        ws->pr[lane + 1] = r;
        if (lane == 0) { ws->pl[0] = 0; ws->pr[0] = 0; }
    }
    __syncthreads();
    const int TR = ws->pr[ntrip];
    // pass 2: which stoppers swap, where their partners are, K and the cut
    int Kw = 0, minNL = 0x7FFFFFFF, minSR = 0x7FFFFFFF;
    bool foundNL = false, foundSR = false;
    for (int t = wv; t < ntrip; t += NT / 64) {
        const int c0 = lo + t * 64 * RP_EPL;
        const int baseL = ws->pl[t], baseR = ws->pr[t];
        T v[RP_EPL];
        int nl_pos = 0x7FFFFFFF, sr_pos = 0x7FFFFFFF;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            int i = c0 + RP_EPL * lane + k;
            v[k] = a[i < hi ? i : lo];
        }
        bool isL[RP_EPL], isR[RP_EPL];
        int pl = 0, pr = 0;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            bool in = c0 + RP_EPL * lane + k < hi;
            isL[k] = in && stopL(v[k]);
            isR[k] = in && stopR(v[k]);
            unsigned long long mL = __ballot(isL[k]), mR = __ballot(isR[k]);
            pl += __popcll(mL & lt); pr += __popcll(mR & lt);
        }
        int cL = baseL + pl, cR = baseR + pr;   // stoppers strictly before this lane's first element
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            const int i = c0 + RP_EPL * lane + k;
            const int kL = cL + 1, kR = TR - cR;
            const bool swL = isL[k] && (TR - cR - (isR[k] ? 1 : 0)) >= kL;   // R_kL lies strictly right of this element
            const bool swR = isR[k] && cL >= kR;                              // L_kR lies strictly left of this element
            if (swR) rpos[kR - 1] = (uint16_t)(i - lo);
            if (isL[k] && !swL) nl_pos = min(nl_pos, i);
            if (swR) sr_pos = min(sr_pos, i);
            Kw += __popcll(__ballot(swL));
            cL += isL[k] ? 1 : 0;
            cR += isR[k] ? 1 : 0;
        }
        if (!foundNL) {
            unsigned long long mk = __ballot(nl_pos != 0x7FFFFFFF);
            if (mk) { minNL = __builtin_amdgcn_readlane(nl_pos, __ffsll((long long)mk) - 1); foundNL = true; }
        }
        if (!foundSR) {
            unsigned long long mk = __ballot(sr_pos != 0x7FFFFFFF);
            if (mk) { minSR = __builtin_amdgcn_readlane(sr_pos, __ffsll((long long)mk) - 1); foundSR = true; }
        }
    }
    if (lane == 0) { ws->k[wv] = Kw; ws->nl[wv] = minNL; ws->sr[wv] = minSR; }
    __syncthreads();
    int K = 0, cut = 0x7FFFFFFF;
#pragma unroll
    for (int q = 0; q < NT / 64; q++) { K += ws->k[q]; cut = min(cut, min(ws->nl[q], ws->sr[q])); }
    // pass 3: the swaps (a left stopper of rank <= K with the right stopper of the same rank)
    for (int t = wv; t < ntrip; t += NT / 64) {
        const int baseL = ws->pl[t];
        if (baseL >= K) break;
        const int c0 = lo + t * 64 * RP_EPL;
        unsigned long long mL[RP_EPL];
        int pl = 0;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            mL[k] = bl[t * RP_EPL + k];
            pl += __popcll(mL[k] & lt);
        }
        int cL = baseL + pl;
#pragma unroll
        for (int k = 0; k < RP_EPL; k++) {
            if ((mL[k] >> lane) & 1ull) {
                const int kL = cL + 1;
                if (kL <= K) {
                    int p = c0 + RP_EPL * lane + k, q = lo + rpos[kL - 1];
                    T vp = a[p], vq = a[q];
                    a[p] = vq;
                    a[q] = vp;
                }
                cL++;
            }
        }
    }
    __syncthreads();
    if (total_r) *total_r = TR;
    return cut;
}

// one partition step for the whole workgroup: cooperative when the range is long enough, else wavefront 0 alone
template <int NT, class T, class P, class FL, class FR>
__device__ __forceinline__ int wg_partition_step(P a, int lo, int hi, FL stopL, FR stopR, uint16_t* rpos, unsigned long long* bl,
                                                 int tid, WgScratch<NT>* ws, int* total_r) {
    const int n = hi - lo;
    if (n >= WG_PARTITION_MIN && n <= WG_PARTITION_MAX) return wg_pair_partition<NT, T>(a, lo, hi, stopL, stopR, rpos, bl, tid, ws, total_r);
    if (tid < 64) {
        int tr = 0;
        int c = wave_pair_partition<T>(a, lo, hi, stopL, stopR, rpos, bl, tid, &tr);
        if (tid == 0) { ws->cut = c; ws->k[0] = tr; }
    }
    __syncthreads();
    const int c = ws->cut;
    if (total_r) *total_r = ws->k[0];
    __syncthreads();  // ws is reused by the next step
    return c;
}

template <int NT, class T, class P>
__device__ __forceinline__ void wg_ls_nth_element(P a, int first, int nth, int last, uint16_t* rpos, unsigned long long* bl, int tid,
                                                  WgScratch<NT>* ws) {
    typedef Rec<T> R;
    if (first == last || nth == last) return;
    int depth = (31 - __clz(last - first)) * 2;
    while (last - first > 3) {
        // below the cooperative partition's range the REST of the selection runs on wavefront 0 alone (its own partition rounds, then
        // the register-resident tail): the other wavefronts wait at ONE barrier instead of three per round (a round on 65 .. 383 records
        // cost 4.5 k cycles with 16 wavefronts at its barriers, profiles/r04_replay_rounds_single_frame.txt)
        if (last - first < WG_PARTITION_MIN) {
            if (tid < 64) wave_ls_introselect<T>(a, first, nth, last, depth, rpos, bl, tid);
            __syncthreads();
            return;
        }
        if (last - first > 65535 || depth == 0) {
            if (tid == 0) ls_introselect<T>(a, first, nth, last, depth);  // identical continuation, one lane
            __syncthreads();
            return;
        }
        --depth;
        int mid = first + (last - first) / 2;
        if (tid == 0) ls_move_median_to_first<T>(a, first, first + 1, mid, last - 1);
        __syncthreads();
        const T pv = a[first];
        int cut = wg_partition_step<NT, T>(
            a, first + 1, last, [pv](T v) { return !R::gt(v, pv); }, [pv](T v) { return !R::gt(pv, v); }, rpos, bl, tid, ws,
            (int*)nullptr);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (tid == 0) ls_insertion_sort<T>(a, first, last);
    __syncthreads();
}

// retainBest for a workgroup of NT threads, all threads convergent; every thread returns the same count
template <int NT, class T, class P>
__device__ __forceinline__ int wg_retain_best(P a, int n, int n_points, int order, uint16_t* rpos, unsigned long long* bl, int tid,
                                              WgScratch<NT>* ws) {
    typedef Rec<T> R;
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    if (order == MO_ORDER_MSVC) {
        if (tid == 0) ms_nth_element<T>(a, 0, n_points - 1, n);
        __syncthreads();
    } else {
        wg_ls_nth_element<NT, T>(a, 0, n_points - 1, n, rpos, bl, tid, ws);
    }
    const T amb = a[n_points - 1];
    const int tail = n - n_points;
    if (tail < REPLAY_SERIAL_BELOW || tail > 65535) {
        if (tid == 0) ws->cut = partition_ge<T>(a, n_points, n, amb);
        __syncthreads();
        const int keep = ws->cut;
        __syncthreads();
        return n_points + keep;
    }
    int total_true = 0;
    wg_partition_step<NT, T>(
        a, n_points, n, [amb](T v) { return !R::ge(v, amb); }, [amb](T v) { return R::ge(v, amb); }, rpos, bl, tid, ws,
        &total_true);
    return n_points + total_true;
}

}  // namespace replay
