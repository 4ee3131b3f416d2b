// so3_rows.h -- the row-streaming engine every large-batch kernel of libso3proj.so runs on, and the
// operations (K1..K4) plugged into it.  Also instantiated by tools/ubench/k1_anatomy.hip (STAMP = true).
//
// The batch is cut into UNITS of 64 rows (64 x 9 elements: 2304 B for f32, 1152 B for bf16).  Persistent
// waves, no workgroup barrier on the data path: in one ROUND a wave takes NPL consecutive units (lane l
// owns row l of each) from up to three input arrays, computes, and writes up to two output arrays; wave w takes
// rounds w, w+W, w+2W, ...  A round's units travel as one block
//     HBM --buffer_load_dwordx4 nt--> VGPR --ds_write_b128--> LDS --ds_read_b32 (stride 9)--> lane
// and back  lane --ds_write_b32--> LDS --ds_read_b128--> VGPR --buffer_store_dwordx4 nt--> HBM.
// Design points (measurements in DESIGN.md):
//   * NPL = 2 packs two independent matrices into the halves of v_pk_* operands (so3_device.h): a Jacobi
//     sweep is a dependent chain and dependent scalar VALU issues at about half rate.
//   * The NEXT round's loads are in flight during the current round's arithmetic and are waited for BEHIND the
//     round's stores: they are older than the stores, so the wait is vmcnt(#stores), never a drain.
//   * Unit I/O is raw buffer instructions with a per-round descriptor (num_records = the bytes that exist, 0 for
//     a prefetch past the end): partial float4 slots, odd tails and the dangling prefetch are dropped by the
//     hardware range check instead of exec-masked branches, which keeps the compiler's vmcnt bookkeeping exact
//     and costs no traffic.
//   * LDS is private to a wave; DS operations of one wave complete in issue order, so no s_barrier is needed.
//   * Every streamed access is non-temporal: each byte is touched once.
//   * Reductions (loss, sum of angles) stay in a per-lane float64 register until the wave retires; one atomic
//     per WORKGROUP then publishes them (same-address float64 atomics cost ~12 ns each).
#pragma once
#ifndef SO3_HOST_MODEL              // (oracle/kernel_model.cpp compiles the pure per-row operations for the host)
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "so3_device.h"

#ifndef SO3_PARK_CAP1
#define SO3_PARK_CAP1 512           // entries of K1's list of parked hard rows (11 words each)
#endif
#ifndef SO3_PARK_CAP2
#define SO3_PARK_CAP2 256           // entries of the two-input kernels' list of parked hard rows (20 words each)
#endif

namespace so3 {
#ifndef SO3_HOST_MODEL

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int kUnitRows = 64;
constexpr int kRsrcFlags = 0x00020000;           // gfx9 raw buffer, 32-bit data format
// Cache policy of the streamed accesses (each byte is touched once): aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1.
// A plain 36 MB -> 36 MB copy: 14.8 us with the default policy, 13.1 us nt (tools/ubench/copy_ceiling.hip).
// (default, sc0, sc1 and their combinations were swept on one device in round 3: nt / nt stays, docs/history/profiles/r03_k1_engine_experiments.txt)
constexpr int kLoadCpol = 2, kStoreCpol = 2;
constexpr int kStreamNt = 2;                     // the cloud kernels' once-read points

// The lane number, recomputed (v_mbcnt in a volatile asm, which is not hoisted): for the rare paths of the engine's operations.  Their address
// arithmetic is loop-invariant; built on the loop's own lane register it was hoisted out of the loop and occupied registers of
// the hot path (K1 at three waves per SIMD: spills).
__device__ __forceinline__ int lane_id_now() {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(static_cast<uint32_t>(b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    const __bf16 h = static_cast<__bf16>(f);   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

// ---- one array's share of a round: geometry and movers ----------------------------------------------------
// EB = element bytes (4: float32, 2: bfloat16, 0: array absent), N = elements per row (9 for 3x3 blocks, 6 for
// the 6D head's input), G = units per round (NPL).  The G units of a round are neighbours in memory and travel as ONE
// block: for G = 2 that is 4608 B = 4.5 KiB of float32 (5 loads per lane instead of 2 x 3) or 2304 B of bfloat16
// (3 loads instead of 2 x 2, all but the last with every lane busy).  An odd N keeps the stride-N LDS accesses
// conflict-free; N = 6 is 2-way conflicted.
template <int EB, int N = 9, int G = 1> struct UnitIO {
    static constexpr int kUnitBytes = kUnitRows * N * EB;        // 2304 (f32 x 9) / 1152 (bf16 x 9) / 1536 (f32 x 6)
    static constexpr int kBytes = G * kUnitBytes;                // the round's block
    static constexpr int kVec4 = kBytes / 16;
    static constexpr int kLoads = (kVec4 + 63) / 64;             // float4 per lane (the last one partial)
    // The LDS image is ALWAYS float32 (bfloat16 is converted once per block on its way in or out): lanes then read their
    // rows with ds_read_b32 at a 9-dword stride, conflict-free, whatever the storage type.  (Sub-dword reads of a bf16
    // image at an 18-byte stride made K1 with bf16 input slower than with float32 input while moving half the bytes.)
    static constexpr int kUnitImage = kUnitRows * N * 4;         // float32 image of one unit; unit k of the round starts at k * kUnitImage
    static constexpr int kSlotBytes = kLoads * 64 * 16 * (4 / EB);   // image of the block incl. the partial load's padding
    static_assert(kBytes % 16 == 0, "a round's block must be a whole number of float4");

    // `unit` (the round's first) and `count` (how many of its G units exist: 0 past the end, G - 1 for an odd tail) are
    // wave-uniform (SGPRs) by construction; accesses beyond count units are dropped by the hardware range check
    static __device__ __forceinline__ rsrc_t rsrc(const void *base, int64_t unit, int count) {
        char *p = static_cast<char *>(const_cast<void *>(base)) + unit * kUnitBytes;
        return __builtin_amdgcn_make_buffer_rsrc(p, 0, count * kUnitBytes, kRsrcFlags);
    }
    static __device__ __forceinline__ void fetch(f32x4 (&v)[kLoads], rsrc_t rs, int lane) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j)
            v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64 * j) * 16, 0, kLoadCpol));
    }
    static __device__ __forceinline__ void store(rsrc_t rs, const f32x4 (&v)[kLoads], int lane) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), rs, (lane + 64 * j) * 16, 0, kStoreCpol);
    }
    // registers (storage type, 16 B per lane and load) -> float32 image in LDS
    static __device__ __forceinline__ void to_lds(char *slot, const f32x4 (&v)[kLoads], int lane) {
        f32x4 *t4 = reinterpret_cast<f32x4 *>(slot);
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            if constexpr (EB == 4) {
                t4[lane + 64 * j] = v[j];
            } else {                                             // 8 bfloat16 -> 8 float32: two 16-byte stores
                const u32x4 w = __builtin_bit_cast(u32x4, v[j]);
                const u32x4 lo = {w.x << 16, w.x & 0xFFFF0000u, w.y << 16, w.y & 0xFFFF0000u};
                const u32x4 hi = {w.z << 16, w.z & 0xFFFF0000u, w.w << 16, w.w & 0xFFFF0000u};
                t4[2 * (lane + 64 * j)] = __builtin_bit_cast(f32x4, lo);
                t4[2 * (lane + 64 * j) + 1] = __builtin_bit_cast(f32x4, hi);
            }
        }
    }
    // float32 image in LDS -> registers in the storage type
    static __device__ __forceinline__ void from_lds(f32x4 (&v)[kLoads], const char *slot, int lane) {
        const f32x4 *t4 = reinterpret_cast<const f32x4 *>(slot);
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            if constexpr (EB == 4) {
                v[j] = t4[lane + 64 * j];
            } else {
                const f32x4 lo = t4[2 * (lane + 64 * j)], hi = t4[2 * (lane + 64 * j) + 1];
                const u32x4 w = {static_cast<uint32_t>(f32_to_bf16_bits(lo.x)) | static_cast<uint32_t>(f32_to_bf16_bits(lo.y)) << 16,
                                 static_cast<uint32_t>(f32_to_bf16_bits(lo.z)) | static_cast<uint32_t>(f32_to_bf16_bits(lo.w)) << 16,
                                 static_cast<uint32_t>(f32_to_bf16_bits(hi.x)) | static_cast<uint32_t>(f32_to_bf16_bits(hi.y)) << 16,
                                 static_cast<uint32_t>(f32_to_bf16_bits(hi.z)) | static_cast<uint32_t>(f32_to_bf16_bits(hi.w)) << 16};
                v[j] = __builtin_bit_cast(f32x4, w);
            }
        }
    }
    // the lane's row of unit `u` of the round <-> LDS image, as component k of T
    template <class T> static __device__ __forceinline__ void read_row(const char *slot, int u, int lane, int k, T (&m)[N]) {
        const float *img = reinterpret_cast<const float *>(slot + u * kUnitImage);
#pragma unroll
        for (int i = 0; i < N; ++i) Tr<T>::set(m[i], k, img[lane * N + i]);
    }
    // the lane's rows of ALL the round's units.  For the packed engine (two units of nine-element rows: K1, K2, K3, K1+K4) that is
    // eighteen ds_read_b32 with immediate offsets off ONE address register, each straight into its half of a register pair, and
    // one wait behind them -- spelled out, because the compiler pairs NEIGHBOURING elements of one row into ds_read2_b32 (it sorts a
    // base's reads by offset) and then moves every dword into its pair: 18 v_mov per round and array, 4 % of K1's vector instructions.
    // (Its waitcnt bookkeeping does not see into an asm statement, hence the wait inside.)
    // The phantom second unit of an odd tail reads its own image: zeros, since the range check returned zeros for it.
    template <class T> static __device__ __forceinline__ void read_rows(const char *slot, int lane, T (&m)[N]) {
        if constexpr (N == 9 && G == 2 && Tr<T>::kLanes == 2) {
            static_assert(kUnitImage == 2304, "the offsets below");
            const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(slot)) + static_cast<unsigned>(lane) * 36u;   // (low word of a flat LDS address = the LDS address)
            float a0, a1, a2, a3, a4, a5, a6, a7, a8, b0, b1, b2, b3, b4, b5, b6, b7, b8;
            asm volatile("ds_read_b32 %0, %18\n\tds_read_b32 %9, %18 offset:2304\n\t"
                         "ds_read_b32 %1, %18 offset:4\n\tds_read_b32 %10, %18 offset:2308\n\t"
                         "ds_read_b32 %2, %18 offset:8\n\tds_read_b32 %11, %18 offset:2312\n\t"
                         "ds_read_b32 %3, %18 offset:12\n\tds_read_b32 %12, %18 offset:2316\n\t"
                         "ds_read_b32 %4, %18 offset:16\n\tds_read_b32 %13, %18 offset:2320\n\t"
                         "ds_read_b32 %5, %18 offset:20\n\tds_read_b32 %14, %18 offset:2324\n\t"
                         "ds_read_b32 %6, %18 offset:24\n\tds_read_b32 %15, %18 offset:2328\n\t"
                         "ds_read_b32 %7, %18 offset:28\n\tds_read_b32 %16, %18 offset:2332\n\t"
                         "ds_read_b32 %8, %18 offset:32\n\tds_read_b32 %17, %18 offset:2336\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7), "=&v"(a8),
                           "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7), "=&v"(b8)
                         : "v"(addr)
                         : "memory");
            m[0] = T{a0, b0}; m[1] = T{a1, b1}; m[2] = T{a2, b2}; m[3] = T{a3, b3}; m[4] = T{a4, b4};
            m[5] = T{a5, b5}; m[6] = T{a6, b6}; m[7] = T{a7, b7}; m[8] = T{a8, b8};
        } else {
            const float *img = reinterpret_cast<const float *>(slot);
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int k = 0; k < G; ++k) Tr<T>::set(m[i], k, img[k * (kUnitImage / 4) + lane * N + i]);
        }
    }
    template <class T> static __device__ __forceinline__ void write_row(char *slot, int u, int lane, int k, const T (&m)[N]) {
        float *img = reinterpret_cast<float *>(slot + u * kUnitImage);
#pragma unroll
        for (int i = 0; i < N; ++i) img[lane * N + i] = Tr<T>::get(m[i], k);
    }
};
template <int N, int G> struct UnitIO<0, N, G> {
    static constexpr int kBytes = 0, kVec4 = 0, kLoads = 1, kSlotBytes = 0;
};

// Per-row side outputs (flip flags, angles): one element per row, contiguous across the lanes of a unit.
template <int BYTES> __device__ __forceinline__ rsrc_t row_rsrc(void *base, int64_t unit, bool exists) {
    char *p = static_cast<char *>(base) + unit * (kUnitRows * BYTES);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, exists ? kUnitRows * BYTES : 0, kRsrcFlags);
}

// ---- reductions without a zero-fill launch and without atomics on the result ----------------------------------
// A caller-owned WORKSPACE (so3_reduce_workspace_bytes(), zero-filled once, used by one stream at a time): every workgroup stores
// its partial into its own slot (performed at the memory side, coherent across the XCDs), takes a ticket, and the workgroup that
// draws the last ticket sums the slots in a fixed order, writes the result and leaves slots, flag and ticket zeroed for the next call.  Against round 2 (a memset or
// 1-thread init launch in front of the kernel, one same-address atomic per workgroup behind it) that is one launch less per
// call, and the sum no longer depends on the order in which workgroups retire: the same input gives the same bits.
constexpr int kMaxPartials = 4094;
struct ReduceWs {
    unsigned int ticket;
    int flag;
    unsigned int pad[2];
    double part[kMaxPartials];
};
static_assert(sizeof(ReduceWs) == 32768, "so3_reduce_workspace_bytes()");

__device__ __forceinline__ double coherent_f64(const double *p) {         // other workgroups' atomics, read past the L2 of this XCD
    return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// A slot holds its partial's bit pattern PLUS ONE: 0, what a slot holds when nobody has written it, then stands for "not here yet" -- no sum of
// angles or norms is the NaN 0xFFFF...F -- and the workgroup that sums the slots can tell a partial that has not arrived from a partial of 0.0.
__device__ __forceinline__ unsigned long long slot_encode(double v) { return static_cast<unsigned long long>(__double_as_longlong(v)) + 1ull; }
__device__ __forceinline__ double slot_decode(unsigned long long s) { return __longlong_as_double(static_cast<long long>(s - 1ull)); }
__device__ __forceinline__ void slot_publish(ReduceWs *ws, unsigned slot, double v) {           // performed at the memory side (agent scope), not waited for
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&ws->part[slot]), slot_encode(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr unsigned kTicketFlag = 0x10000u;         // a workgroup whose rows raised the flag draws its ticket with this on top (counts stay below 65 536)
#ifndef SO3_SLOT_POLLS
#define SO3_SLOT_POLLS 200000                      // ~0.1 s of polling one slot before the sum is declared lost (NaN): a store issued in front of a ticket arrives within microseconds
#endif

// Called by ALL threads of a workgroup at the end of the kernel; wg_total / wg_flag are thread 0's.  `expected` = how many
// workgroups take tickets in this launch (0: publish only -- a remainder kernel whose partials the following launch collects),
// `npart` = slots to sum.  write_result(total, any_flag) runs on thread 0 of the last workgroup.
// Round 6: ONE round trip to memory per workgroup instead of two.  Rounds 3-5 added the partial to its slot with a RETURNING atomic, waited for
// it (so that it had been performed), and only then drew the ticket -- two dependent memory-side operations of ~1 us each behind every
// workgroup's last row, and a third (the slots' loads) behind the last one.  Now the partial is a plain agent-scope store that nobody waits for
// and the slots synchronise themselves: the summing workgroup polls a slot that still reads "not here yet" (rare: the store was issued in front
// of the ticket it has just seen).  The flag rides on the ticket.  Same slots, same order of summation, same bits from call to call.
template <int BLOCK, class F>
__device__ __forceinline__ void ticket_finish(ReduceWs *ws, unsigned slot, unsigned expected, unsigned npart, double wg_total, bool wg_flag,
                                              F &&write_result) {
    __shared__ int is_last;
    __shared__ unsigned ticket_flags;
    __shared__ int lost;
    __shared__ double fin[BLOCK / 64];
    if (threadIdx.x == 0) {
        // Everything that crosses workgroups here is performed at the memory side (agent scope, past the XCDs' L2s) or a coherent load: no
        // release fence -- a __threadfence() per workgroup writes back the XCD's whole L2, i.e. the kernel's own streamed output, and made K3
        // 50 us instead of 31.
        slot_publish(ws, slot, wg_total);
        int last = 0;
        unsigned flags = wg_flag ? 1u : 0u;
        if (expected != 0) {
            const unsigned seen = atomicAdd(&ws->ticket, 1u + (wg_flag ? kTicketFlag : 0u));
            last = (seen & (kTicketFlag - 1u)) == expected - 1 ? 1 : 0;
            flags += seen >> 16;
        } else if (wg_flag) {
            atomicOr(&ws->flag, 1);                            // publish only: the launch that collects comes later on the stream
        }
        is_last = last;
        ticket_flags = flags;
        lost = 0;
    }
    __syncthreads();
    if (!is_last) return;
    double v = 0.0;
    for (unsigned i = threadIdx.x; i < npart; i += BLOCK) {            // thread t sums slots t, t + BLOCK, ...: a fixed order
        unsigned long long *p = reinterpret_cast<unsigned long long *>(&ws->part[i]);
        unsigned long long got = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int polls = 0; got == 0ull && polls < SO3_SLOT_POLLS; ++polls) {
            __builtin_amdgcn_s_sleep(2);
            got = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (got == 0ull) lost = 1;                                     // (never seen; the result then says so instead of being a wrong number)
        else v += slot_decode(got);
        __hip_atomic_store(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) fin[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) total += fin[w];
        if (lost) total = __longlong_as_double(0x7ff8000000000000ll);
        const int flag = __hip_atomic_load(&ws->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // set by a remainder kernel of an earlier launch
        write_result(total, flag != 0 || ticket_flags != 0u);
        __hip_atomic_store(&ws->flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#endif  // !SO3_HOST_MODEL

// A reservation of n consecutive entries in a list of `cap` (the workgroup's list of parked hard rows, below): -1 when they do
// not fit.  A compare-and-swap, so an attempt that fails never touches the count: round 4 added n and took it back on failure, and
// with three waves racing near a full list a reservation could succeed BETWEEN another's add and its subtract -- its base then
// counted entries nobody wrote, and the redo pass would have stored through row numbers read from uninitialised LDS.  Generic
// over the primitive (cas(p, expected, desired) returns what it found; load(p) reads the count) so that the CPU suite replays
// interleavings on it (oracle/kernel_model.cpp: model_park_reserve_interleaved).
template <class Load, class Cas>
__device__ __forceinline__ int park_reserve_protocol(unsigned *count, unsigned n, unsigned cap, Load &&load, Cas &&cas) {
    unsigned old = load(count);
    while (true) {
        if (old + n > cap) return -1;
        const unsigned seen = cas(count, old, old + n);
        if (seen == old) return static_cast<int>(old);
        old = seen;
    }
}

template <int NPL> struct LaneT;
template <> struct LaneT<1> { typedef float type; };
template <> struct LaneT<2> { typedef f32x2 type; };

// The rows of the current round as an operation sees them: up to three inputs, up to two outputs.
template <class T, class Op> struct Rows {
    T a[Op::kIn0N], b[Op::kIn1N], c[Op::kIn2N];
    T o0[Op::kOut0N], o1[Op::kOut1N];
};

// What an operation sees of the current round.
template <int NPL> struct RowCtx {
    int64_t unit[NPL];      // unit index of component k (wave-uniform)
    bool exists[NPL];       // false only for the phantom second unit of an odd tail
    int lane;
    double acc;             // per-lane float64 partial of the operation's reduction
    bool flag;              // per-lane sticky flag (K4: cosine out of range)
    const char *img1;       // LDS image of the second input's block (operations with kLateIn1 read their rows from it themselves)
    char *slot;             // the wave's LDS slot: free between take_rows and put_rows (the inputs have left, the outputs are not staged yet)
    int dense;              // wave-uniform (K1): 1 = the wave's last round was dense in hard rows, 2 = so were earlier ones and the shortcut was refused
    float *park;            // Op::kParkWords > 0: the workgroup's list of parked hard rows (LDS, see park_hard_rows) ...
    unsigned *park_count;   // ... and how many it holds
    int n_half, n_pi;       // wave-uniform (the float32 angle sum, angle_sum_f32): rows of this wave whose angle is pi/2 - r, pi - 2r
};

#ifndef SO3_HOST_MODEL
// ---- the engine ----------------------------------------------------------------------------------------
// Op provides: kIn0, kIn1, kIn2, kOut0, kOut1 (element bytes, 0 = absent), pointers in0, in1, in2, out0, out1,
//   kIn0N .. kOut1N (elements per row, 9 unless overridden),
//   template <class T, int NPL> void compute(Rows<T, Op> &rows, RowCtx<NPL> &)      reads rows.a/b/c, writes rows.o0/o1
//   void finish(double block_total, bool any_flag)      -- called by thread 0 of each workgroup at the end
//   kReduce: whether acc/flag are used.
// WPS = resident waves per SIMD the register budget is sized for (the host launches CUs * 4 * WPS * 64 / BLOCK
// workgroups); wave w of the grid takes rounds w, w + W, w + 2W, ... (static deal: tickets, two rounds in flight and
// s_setprio were measured in round 2 and bought nothing).
// (Inputs straight into LDS -- buffer_load_dwordx4 ... lds, two images per wave, no VGPR staging -- was built and measured in
// round 3: the copy through the engine 2 % faster, K1 5-14 % SLOWER, K2 / K3 no better; it lives in the history, not here.)
// STAMP (diagnostic builds only, instantiated by tools/ubench/k1_anatomy.hip from its own translation unit): per wave
//   {s_memrealtime entry, exit, s_memtime entry, cycles | XCC << 28 | HW_ID << 32, rounds done << 48, 0} and the
//   begin / end of the arithmetic of the wave's first four rounds.
template <class Op> struct EngineIO {
    template <int NPL> struct For {
        typedef UnitIO<Op::kIn0, Op::kIn0N, NPL> I0;
        typedef UnitIO<Op::kIn1, Op::kIn1N, NPL> I1;
        typedef UnitIO<Op::kIn2, Op::kIn2N, NPL> I2;
        typedef UnitIO<Op::kOut0, Op::kOut0N, NPL> O0;
        typedef UnitIO<Op::kOut1, Op::kOut1N, NPL> O1;
    };
};

template <class Op, int NPL, int WPS, int BLOCK, bool STAMP = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void k_rows(Op op, int64_t nunits, unsigned long long *__restrict__ stamps) {
    typedef typename LaneT<NPL>::type T;
    typedef typename EngineIO<Op>::template For<NPL> IO;
    typedef typename IO::I0 I0;
    typedef typename IO::I1 I1;
    typedef typename IO::I2 I2;
    typedef typename IO::O0 O0;
    typedef typename IO::O1 O1;
    constexpr int kWaves = BLOCK / 64;
    // LDS slot of a wave: the images of the round's input blocks side by side; outputs are staged over them once the
    // rows are in registers.  Images are padded to whole 1-KiB loads.
    constexpr int kIn0B = I0::kSlotBytes, kIn1B = I1::kSlotBytes, kIn2B = I2::kSlotBytes;
    constexpr int kOut0B = O0::kSlotBytes, kOut1B = O1::kSlotBytes;
    constexpr int kInBytes = kIn0B + kIn1B + kIn2B;
    constexpr int kOutBytes = kOut0B + kOut1B;
    constexpr int kSlot = kInBytes > kOutBytes ? kInBytes : kOutBytes;
    __shared__ __attribute__((aligned(16))) char lds[kWaves][kSlot];
    // Op::kParkWords > 0: the rows the operation's fast path cannot serve and that are FEW in their round wait here, inputs and
    // row number, word w of entry e at [w * kParkCap + e]; Op::redo_parked runs on them when the workgroup has streamed its share
    __shared__ float park[Op::kParkWords > 0 ? (Op::kParkWords + 2) * Op::kParkCap : 1];
    __shared__ unsigned park_count;
    __shared__ double red[Op::kReduce ? kWaves : 1];
    __shared__ int red_flag[Op::kReduce ? kWaves : 1];

    unsigned long long t_real0 = 0, t_mem0 = 0, rounds_done = 0;
    if (STAMP) { t_real0 = __builtin_amdgcn_s_memrealtime(); t_mem0 = __builtin_amdgcn_s_memtime(); }
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // SGPR: unit indices stay scalar
    // Round numbers are 32-bit: the scalar unit has no 64-bit ordered compare, and as int64 the loop's "any rounds left", "how many units
    // of this round exist" went through v_cmp_*_i64 on the vector unit every round.  (2^31 rounds of 128 rows are 10 TB of float32 rows;
    // the host refuses a batch beyond that, stream_units.)  Units and rows stay 64-bit where they become addresses.
    constexpr int kFixed = Op::kFixedRounds;
    const int nrounds_all = static_cast<int>((nunits + NPL - 1) / NPL);
    const int stride = kFixed > 0 ? 1 : static_cast<int>(gridDim.x) * kWaves;
    const int wave_id = static_cast<int>(blockIdx.x) * kWaves + wave_in_block;
    int t = kFixed > 0 ? wave_id * kFixed : wave_id;
    // the wave's rounds are t, t + stride, ... below `nrounds`: all of the launch's for a persistent wave, its own k for a fixed one
    const int nrounds = kFixed > 0 ? (t + kFixed < nrounds_all ? t + kFixed : nrounds_all) : nrounds_all;
    RowCtx<NPL> ctx;
    ctx.lane = lane;
    ctx.acc = 0.0;
    ctx.flag = false;
    ctx.img1 = nullptr;
    ctx.slot = nullptr;
    ctx.dense = 0;
    ctx.park = park;
    ctx.park_count = &park_count;
    if constexpr (Op::kParkWords > 0) {
        if (threadIdx.x == 0) park_count = 0;
        __syncthreads();
    }
    ctx.n_half = 0;
    ctx.n_pi = 0;
    if (t < nrounds) {
        const int nunits32 = static_cast<int>(nunits);
        auto units_of = [&](int tr) -> int {                // how many of round tr's NPL units exist (0 past the end)
            const int left = nunits32 - tr * NPL;
            return tr < nrounds ? (left < NPL ? left : NPL) : 0;
        };
        // STAMP builds: wall-clock (100 MHz) begin / end of the arithmetic of the wave's first four rounds
        auto phase = [&](int ph) {
            if (STAMP) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (rounds_done < 4 && lane == 0)
                    stamps[6 * static_cast<int64_t>(gridDim.x) * kWaves + 40 * static_cast<int64_t>(wave_id) + 10 * rounds_done + ph] = now;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // the lanes' rows of round t out of the wave's image `img`
        auto take_rows = [&](const char *img, Rows<T, Op> &rows) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                ctx.unit[k] = t * NPL + k;
                ctx.exists[k] = t * NPL + k < nunits32; // wave-uniform; false only for the phantom unit of an odd tail, whose lanes
            }                                           // work on the zeros the range check returned for it (results dropped)
            I0::template read_rows<T>(img, lane, rows.a);
            if constexpr (Op::kIn1 != 0 && !Op::kLateIn1) I1::template read_rows<T>(img + kIn0B, lane, rows.b);
            if constexpr (Op::kIn2 != 0) I2::template read_rows<T>(img + kIn0B + kIn1B, lane, rows.c);
        };
        // the round's results: rows -> image (over the consumed inputs) -> float4 per lane -> HBM
        auto put_rows = [&](char *img, const Rows<T, Op> &rows) {
            if constexpr (Op::kOut0 != 0 || Op::kOut1 != 0) {
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    if constexpr (Op::kOut0 != 0) O0::write_row(img, k, lane, k, rows.o0);
                    if constexpr (Op::kOut1 != 0) O1::write_row(img + kOut0B, k, lane, k, rows.o1);
                }
                wave_lds_fence();
                f32x4 v0[O0::kLoads], v1[O1::kLoads];
                if constexpr (Op::kOut0 != 0) O0::from_lds(v0, img, lane);
                if constexpr (Op::kOut1 != 0) O1::from_lds(v1, img + kOut0B, lane);
                wave_lds_fence();
                const int cnt = units_of(t);            // an odd tail's phantom unit is cut off by the descriptor
                if constexpr (Op::kOut0 != 0) O0::store(O0::rsrc(op.out0, static_cast<int64_t>(t) * NPL, cnt), v0, lane);
                if constexpr (Op::kOut1 != 0) O1::store(O1::rsrc(op.out1, static_cast<int64_t>(t) * NPL, cnt), v1, lane);
            }
        };
        {
            char *slot = lds[wave_in_block];
            ctx.img1 = slot + kIn0B;
            ctx.slot = slot;
            // One round is in flight in registers behind the round that sits in LDS.
            f32x4 in0[I0::kLoads], in1[I1::kLoads], in2[I2::kLoads];
            auto issue = [&](int tr) {                      // past the last round the descriptors are empty: the loads
                const int cnt = units_of(tr);               // return 0 and cost no traffic
                const int64_t u = static_cast<int64_t>(tr) * NPL;      // (past the end nothing is in range, whatever the base)
                I0::fetch(in0, I0::rsrc(op.in0, u, cnt), lane);
                if constexpr (Op::kIn1 != 0) I1::fetch(in1, I1::rsrc(op.in1, u, cnt), lane);
                if constexpr (Op::kIn2 != 0) I2::fetch(in2, I2::rsrc(op.in2, u, cnt), lane);
            };
            auto land = [&]() {                             // registers -> the wave's LDS slot (float32 images)
                I0::to_lds(slot, in0, lane);
                if constexpr (Op::kIn1 != 0) I1::to_lds(slot + kIn0B, in1, lane);
                if constexpr (Op::kIn2 != 0) I2::to_lds(slot + kIn0B + kIn1B, in2, lane);
            };
            issue(t);
            land();
            int held = t + stride;                          // the round the registers hold (>= nrounds: none, empty loads)
            issue(held);
            while (true) {
                wave_lds_fence();
                Rows<T, Op> rows;
                take_rows(slot, rows);
                wave_lds_fence();
                phase(3);
                op.template compute<T, NPL>(rows, ctx);
                phase(4);
                if (STAMP) ++rounds_done;
                put_rows(slot, rows);
                if (held >= nrounds) break;
                // The prefetched round lands in LDS here, at the END of the body: its loads are older than this round's
                // stores, so the wait is vmcnt(#stores), never a drain.
                land();
                t = held;
                held += stride;
                issue(held);
            }
        }
    }
    if constexpr (Op::kParkWords > 0) {
        // The workgroup has streamed its share: its parked rows are redone, 64 per wave and pass, one matrix per lane
        // (Op::redo_parked), and their outputs overwrite what the rounds' block stores wrote for them -- every wave's stores have
        // been performed by then, and one L2 serves the whole workgroup, so the later store lands on top.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned parked = park_count;
        const int total = parked < static_cast<unsigned>(Op::kParkCap) ? static_cast<int>(parked) : Op::kParkCap;
        // Pass p goes to wave (p + first) mod kWaves, `first` differing between the workgroups that share a CU (consecutive ones,
        // or ones 256 apart): wave w of every resident workgroup sits on SIMD w, and with a few parked rows per workgroup -- one
        // pass each -- wave 0 of all of them would queue on SIMD 0 while the other three SIMDs idle.
        const int first = static_cast<int>((blockIdx.x + (blockIdx.x >> 8)) % kWaves);
        for (int e0 = kUnitRows * ((wave_in_block + kWaves - first) % kWaves); e0 < total; e0 += kUnitRows * kWaves) {
            const int ln = lane_id_now();
            const bool valid = e0 + ln < total;
            op.template redo_parked<NPL>(ctx, valid ? e0 + ln : e0, valid);
        }
    }
    if (Op::kReduce) {
        double v = ctx.acc;
        if constexpr (Op::kAngleConstants)        // angle_sum_f32: the multiples of pi/2 the wave's rows carry, counted on the scalar unit
            if (lane == 0) v += 1.57079632679489661923 * static_cast<double>(ctx.n_half) + 3.14159265358979323846 * static_cast<double>(ctx.n_pi);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        const bool any_flag = __any(ctx.flag);
        if (lane == 0) { red[wave_in_block] = v; red_flag[wave_in_block] = any_flag ? 1 : 0; }
        __syncthreads();
        double total = 0.0;
        int f = 0;
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) { total += red[w]; f |= red_flag[w]; }
            total = op.scale_partial(total);
        }
        if (op.ws == nullptr) {                             // no workspace: atomics onto accumulators the host initialised
            if (threadIdx.x == 0) op.finish(total, f != 0);
        } else {
            ticket_finish<BLOCK>(op.ws, op.ws_slot0 + blockIdx.x, gridDim.x, op.ws_slot0 + gridDim.x, total, f != 0,
                                 [&](double t, bool any) { op.finish_total(t, any); });
        }
    }
    if (STAMP && (kFixed > 0 ? wave_id * kFixed : wave_id) < nrounds_all) {
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * kWaves + wave_in_block;
            stamps[6 * wave_id + 0] = t_real0;
            stamps[6 * wave_id + 1] = __builtin_amdgcn_s_memrealtime();
            stamps[6 * wave_id + 2] = t_mem0;
            // HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20): which XCD / SE / CU / SIMD ran this wave
            const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
            stamps[6 * wave_id + 3] = ((__builtin_amdgcn_s_memtime() - t_mem0) & 0xFFFFFFFull)
                                      | (static_cast<unsigned long long>(xcc & 0xF) << 28) | (static_cast<unsigned long long>(hw) << 32);
            stamps[6 * wave_id + 4] = rounds_done << 48;
            stamps[6 * wave_id + 5] = 0;
        }
    }
}

#endif  // !SO3_HOST_MODEL

#ifndef SO3_HOST_MODEL
// The rows of the second input for an operation with kLateIn1 (called from its compute(), before the results are staged).
template <class T, class Op, int NPL> __device__ __forceinline__ void late_in1(const RowCtx<NPL> &ctx, T (&b)[Op::kIn1N]) {
    typedef UnitIO<Op::kIn1, Op::kIn1N, NPL> I1;
    I1::template read_rows<T>(ctx.img1, ctx.lane, b);
}
#endif

// ---- the operations --------------------------------------------------------------------------------------
struct OpBase {
    static constexpr int kIn2 = 0;                                         // most operations have at most two inputs
    static constexpr int kIn0N = 9, kIn1N = 9, kIn2N = 9, kOut0N = 9, kOut1N = 9;     // elements per row of each array
    const void *in0 = nullptr, *in1 = nullptr, *in2 = nullptr;
    void *out0 = nullptr, *out1 = nullptr;
    static constexpr bool kReduce = false;
    static constexpr bool kAngleConstants = false;      // the operation sums angles through angle_sum_f32 (RowCtx::n_half, n_pi)
    // kLateIn1: the operation reads the second input's rows out of LDS itself (late_in1), where it first needs them -- for K2 / K3 /
    // K1+K4 that is AFTER the projection, whose ~110 live registers the 18 of a second input's rows would otherwise sit beside
    static constexpr bool kLateIn1 = false;
    // kParkWords > 0: the operation parks the inputs of the few hard rows of a round (park_hard_rows: kParkWords words per row, at
    // most kParkCap rows per workgroup) and provides
    //   template <int NPL> void redo_parked(RowCtx<NPL> &, int entry, bool valid)     one row per lane, behind the loop
    static constexpr int kParkWords = 0, kParkCap = 1;
    // kFixedRounds > 0 (experiment builds, SO3_K1_FIXED_ROUNDS): NOT persistent -- wave w takes the kFixedRounds CONSECUTIVE rounds
    // w k, w k + 1, ... and retires; the host launches ceil(rounds / k) waves and the dispatcher back-fills (round 6's A/B)
    static constexpr int kFixedRounds = 0;
#ifndef SO3_HOST_MODEL
    ReduceWs *ws = nullptr;        // reduction workspace (nullptr: atomics onto host-initialised accumulators)
    unsigned ws_slot0 = 0;         // slots below this one were filled by the remainder kernel launched before the engine
#endif
    __device__ __forceinline__ void finish(double, bool) const {}
    __device__ __forceinline__ void finish_total(double, bool) const {}
    __device__ __forceinline__ double scale_partial(double t) const { return t; }      // a workgroup's partial -> the unit of the result
};

#ifndef SO3_HOST_MODEL   // K1..K4 write side outputs through buffer descriptors and publish reductions with atomics
// ---- rows the fast path cannot serve ------------------------------------------------------------------------------------
// The quaternion fast path (so3_device.h section 3a) declares a row HARD when it cannot certify its rotation: 2e-6 of Gaussian
// rows, every row of a batch of reflections, ties, rank-deficient or zero matrices.  SIMT leaves no way to run the second
// algorithm on the hard lanes alone at less than full price, so:
//   * a round with FEW hard rows (at most 16 NPL of its 64 NPL) PARKS them -- inputs and row number, in a list the workgroup
//     shares in LDS -- and streams on; the block store writes whatever the fast path left for them.  When the workgroup has
//     streamed its share the list is redone, 64 rows per wave and pass, ONE matrix per lane (signed_svd<., float>: the same IEEE
//     operations per matrix as the packed instantiation, so a row's bits do not depend on where it waited), and the results
//     overwrite the rows.  The cost is per hard ROW: round 3 queued per wave and ran one packed Jacobi pass per wave that held any
//     (K1; 1 % of hard rows: 1.3-1.46 x a Gaussian batch), K2 / K3 / K1+K4 paid a packed pass per ROUND that held one.
//   * a round DENSE in hard rows (or one the list has no room for) runs the packed Jacobi path on the spot, in front of the block
//     store, as before: parked, such rows would be stored twice, the second time four bytes per lane and store.
// (Built and measured first: row NUMBERS parked instead of inputs, every hard row redone from memory behind the loop, no Jacobi
// code in any loop -- docs/history/profiles/r04_row_number_queue*: the redo waits for its re-read inputs and for the round's stores, 10 % of
// hard rows 1.8 x, whole batches 2.0-2.3 x.  The same experiment showed K2 at three waves per SIMD, spill-free, no faster than at two.)
template <int CAP> __device__ __forceinline__ int park_reserve(unsigned *count, int n) {
    int base = 0;
    if (lane_id_now() == 0)
        base = park_reserve_protocol(count, static_cast<unsigned>(n), static_cast<unsigned>(CAP),
                                     [](unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); },
                                     [](unsigned *p, unsigned expected, unsigned desired) { return atomicCAS(p, expected, desired); });
    return __builtin_amdgcn_readfirstlane(base);
}
// How many hard rows the round holds (wave-uniform), and the list entry of each hard lane-half given the round's first entry.
template <class T, int NPL>
__device__ __forceinline__ int count_hard(const RowCtx<NPL> &ctx, typename Tr<T>::mask hard) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < NPL; ++k) n += __builtin_popcountll(__builtin_amdgcn_ballot_w64(Tr<T>::lane_of(hard, k) && ctx.exists[k]));
    return n;
}
// words [w0, w0 + N) of the round's parked rows <- v;  ROW: and the row number (two words behind the operation's WORDS)
template <class T, int NPL, int CAP, int WORDS, bool ROW, int N>
__device__ __forceinline__ void park_words(const RowCtx<NPL> &ctx, int base, typename Tr<T>::mask hard, int w0, const T (&v)[N]) {
    const int lane = lane_id_now();
    int at = base;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const bool mine = Tr<T>::lane_of(hard, k) && ctx.exists[k];
        const unsigned long long votes = __builtin_amdgcn_ballot_w64(mine);
        if (mine) {
            const int e = at + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(votes >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(votes), 0u)));
#pragma unroll
            for (int i = 0; i < N; ++i) ctx.park[(w0 + i) * CAP + e] = Tr<T>::get(v[i], k);
            if (ROW) {
                const long long row = ctx.unit[k] * kUnitRows + lane;
                ctx.park[WORDS * CAP + e] = __int_as_float(static_cast<int>(row & 0xffffffffll));
                ctx.park[(WORDS + 1) * CAP + e] = __int_as_float(static_cast<int>(row >> 32));
            }
        }
        at += __builtin_popcountll(votes);
    }
}
template <int CAP, int WORDS> __device__ __forceinline__ long long parked_row(const float *park, int e) {
    return static_cast<long long>(static_cast<unsigned>(__float_as_int(park[WORDS * CAP + e]))) | (static_cast<long long>(__float_as_int(park[(WORDS + 1) * CAP + e])) << 32);
}
template <int CAP, int N> __device__ __forceinline__ void parked_words(const float *park, int e, int w0, float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = park[(w0 + i) * CAP + e];
}
// One row of an output array by its number, for the redo pass (EB: 4 = float32, 2 = bfloat16).  A float32 row leaves as three
// 12-byte stores with the default cache policy: the row is a PARTIAL write into lines the block store has streamed out already, and
// nine non-temporal dword stores per row (round 3's flush) are nine read-modify-writes at the memory side -- with 10 % of hard
// rows that alone was a third of the launch.
typedef float f32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
template <int EB, int N> __device__ __forceinline__ void overwrite_row(void *base, long long row, const float (&m)[N]) {
    if constexpr (EB == 4 && N % 3 == 0) {
        float *p = static_cast<float *>(base) + row * N;
#pragma unroll
        for (int i = 0; i < N; i += 3) {
            const f32x3_a4 v = {m[i], m[i + 1], m[i + 2]};
            *reinterpret_cast<f32x3_a4 *>(p + i) = v;
        }
    } else if constexpr (EB == 4) {
        float *p = static_cast<float *>(base) + row * N;
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = m[i];
    } else {
        uint16_t *p = static_cast<uint16_t *>(base) + row * N;
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = f32_to_bf16_bits(m[i]);
    }
}
// The decision of a round that holds hard rows: -1 = dense (or no room): the caller runs the Jacobi path on the spot; otherwise the
// round's first entry in the workgroup's list, where the caller parks the rows' inputs and numbers (park_words).
template <class T, int NPL, int CAP>
__device__ __forceinline__ int park_hard_rows(const RowCtx<NPL> &ctx, typename Tr<T>::mask hard, bool *dense = nullptr) {
    const int n = count_hard<T, NPL>(ctx, hard);
    if (dense != nullptr) *dense = n > 16 * NPL;
    if (n > 16 * NPL) return -1;
    return park_reserve<CAP>(ctx.park_count, n);
}

// The Jacobi path's rotation for everything a lane holds, one matrix at a time through the one-matrix-per-lane instantiation (for a
// packed pair: two passes of ~570 plain instructions instead of one of ~570 mostly packed ones, +10 % in cycles) -- where the
// packed body's 150 registers do not fit beside what the loop keeps live (K1 at three waves per SIMD: the packed body spilled 20
// registers around its peak, and a dense batch paid for the scratch traffic).  Same bits either way.
template <class T> __device__ __forceinline__ void jacobi_rotation_by_halves(const T (&m)[9], T (&rj)[9]) {
#pragma unroll
    for (int k = 0; k < Tr<T>::kLanes; ++k) {
        float mk_[9], rk_[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) mk_[i] = Tr<T>::get(m[i], k);
        rotation_from(signed_svd<false, float>(mk_), rk_);
#pragma unroll
        for (int i = 0; i < 9; ++i) Tr<T>::set(rj[i], k, rk_[i]);
    }
}

// K1's arithmetic for a round (so3_device.h sections 3a / 3b) with the round's hard rows parked -- inputs and row numbers -- when they are
// few, or through the Jacobi path on the spot when the round is dense in them.  Returns the PARKED rows (their r is whatever the fast path
// left; Op::redo_parked answers for them).
template <class T, int NPL, int CAP, int WORDS>
__device__ __forceinline__ typename Tr<T>::mask project_or_park(const T (&m)[9], T (&r)[9], RowCtx<NPL> &ctx) {
    typedef Tr<T> R;
    const typename R::mask none = R::gt(R::splat(0.f), R::splat(1.f));
    // After a round dense in hard rows the next one is asked first whether ALL its rows are hard by their invariants alone
    // (a batch of reflections, rank-one or zero rows): then it takes the Jacobi path without running the fast path at all.
    if (__builtin_expect(ctx.dense == 1, 0) && all_rows_invariant_hard<T>(m)) {
        rotation_from(signed_svd<false, T>(m), r);
        return none;
    }
    const int asked = ctx.dense;                     // 1: the question was asked and the answer was no -- a batch of ties, say: not again in this wave
    const typename R::mask hard = quat_rotation<T, false>(m, r);     // no early way out (SKIP): at three waves per SIMD K1 has no registers to spare for it
    ctx.dense = 0;
    if (__builtin_expect(R::wave_any(hard), 0)) {
        bool dense;
        const int base = park_hard_rows<T, NPL, CAP>(ctx, hard, &dense);
        ctx.dense = dense ? (asked != 0 ? 2 : 1) : 0;
        if (base >= 0) {
            park_words<T, NPL, CAP, WORDS, true, 9>(ctx, base, hard, 0, m);
            return hard;
        } else {
            // The fast path's rotations wait in the wave's LDS slot meanwhile (the round's inputs have left it, its outputs are
            // not staged yet): 18 registers that the Jacobi path's peak would otherwise sit on top of.
            typedef UnitIO<4, 9, NPL> Stash;
            const int lane = lane_id_now();
#pragma unroll
            for (int k = 0; k < NPL; ++k) Stash::write_row(ctx.slot, k, lane, k, r);
            wave_lds_fence();                                // (also keeps the compiler from forwarding the stores to the loads below)
            T rj[9];
            jacobi_rotation_by_halves<T>(m, rj);
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < NPL; ++k) Stash::read_row(ctx.slot, k, lane, k, r);
#pragma unroll
            for (int j = 0; j < 9; ++j) r[j] = R::sel(hard, rj[j], r[j]);
        }
    }
    return none;
}

// K1: R = U diag(1,1,det(UV^T)) V^T  (rotation_representation.py:192-206): the quaternion fast path; its hard rows parked, or
// through the packed Jacobi path on the spot when the round is dense in them.
#ifndef SO3_K1_FIXED_ROUNDS
#define SO3_K1_FIXED_ROUNDS 0
#endif
template <int IN_BYTES, bool FLIP>
struct OpProject : OpBase {
    static constexpr int kIn0 = IN_BYTES, kIn1 = 0, kOut0 = 4, kOut1 = 0;
    static constexpr int kFixedRounds = SO3_K1_FIXED_ROUNDS;
    static constexpr int kParkWords = 9, kParkCap = SO3_PARK_CAP1;          // 512 entries: 22 KB of LDS per workgroup
    uint8_t *flip = nullptr;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpProject> &rows, RowCtx<NPL> &ctx) const {
        typedef Tr<T> R;
        const T (&m)[9] = rows.a;
        T (&r)[9] = rows.o0;
        if (FLIP) {                                          // the flags first: their temporaries are gone before the projection's peak
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                float mk_[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) mk_[i] = Tr<T>::get(m[i], k);
                const unsigned char bit = det_negative(mk_) ? 1 : 0;
                __builtin_amdgcn_raw_buffer_store_b8(bit, row_rsrc<1>(flip, ctx.exists[k] ? ctx.unit[k] : 0, ctx.exists[k]), ctx.lane, 0, 0);
            }
        }
        project_or_park<T, NPL, kParkCap, kParkWords>(m, r, ctx);
    }
    template <int NPL>
    __device__ __forceinline__ void redo_parked(RowCtx<NPL> &ctx, int e, bool valid) const {
#pragma clang fp contract(on)         // as the Jacobi path it continues (so3_device.h)
        float m[9], r[9];
        parked_words<kParkCap, 9>(ctx.park, e, 0, m);
        rotation_from(signed_svd<false, float>(m), r);
        if (valid) overwrite_row<4, 9>(out0, parked_row<kParkCap, kParkWords>(ctx.park, e), r);
    }
};

// K2: dM = U' Bm V^T for upstream G (autograd of K1): the fast path's rotation and the backward from the rotation alone (so3_device.h
// section 3c); hard rows through the Jacobi frames and their floored denominators -- parked, or on the spot in a dense round.
template <int M_BYTES>
struct OpProjectBwd : OpBase {
    static constexpr int kIn0 = M_BYTES, kIn1 = 4, kOut0 = M_BYTES, kOut1 = 0;
    static constexpr bool kLateIn1 = true;
    static constexpr int kParkWords = 18, kParkCap = SO3_PARK_CAP2;                   // M and G: 20 KB of LDS per workgroup
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpProjectBwd> &rows, RowCtx<NPL> &ctx) const {
        typedef Tr<T> R;
        T r[9];
        HardRows<T> h;
        h.hard = quat_rotation<T>(rows.a, r, &h.prescale);
        h.any = false;
        int base = -1;
        if (__builtin_expect(R::wave_any(h.hard), 0)) {
            base = park_hard_rows<T, NPL, kParkCap>(ctx, h.hard);
            if (base >= 0) {
                park_words<T, NPL, kParkCap, kParkWords, true, 9>(ctx, base, h.hard, 0, rows.a);
            } else {
                h.any = true;
                h.frames = signed_svd<true, T>(rows.a);
            }
        }
        late_in1<T, OpProjectBwd, NPL>(ctx, rows.b);              // G: only now
        if (__builtin_expect(base >= 0, 0)) park_words<T, NPL, kParkCap, kParkWords, false, 9>(ctx, base, h.hard, 9, rows.b);
        backward_given_rotation<T>(rows.a, r, rows.b, h, rows.o0);
    }
    template <int NPL>
    __device__ __forceinline__ void redo_parked(RowCtx<NPL> &ctx, int e, bool valid) const {
#pragma clang fp contract(on)         // as the Jacobi path it continues (so3_device.h)
        float m[9], g[9], dm[9];
        parked_words<kParkCap, 9>(ctx.park, e, 0, m);
        parked_words<kParkCap, 9>(ctx.park, e, 9, g);
        project_backward(signed_svd<true, float>(m), g, dm);
        if (valid) overwrite_row<M_BYTES, 9>(out0, parked_row<kParkCap, kParkWords>(ctx.park, e), dm);
    }
};

// K3: head + Frobenius loss + backward in one pass (3D-Pose/main.py:60,85,90).  out0 = dM, out1 = R (each optional).
template <int M_BYTES, bool WANT_DM, bool WANT_R>
struct OpFrobHead : OpBase {
    static constexpr int kIn0 = M_BYTES, kIn1 = 4, kOut0 = WANT_DM ? M_BYTES : 0, kOut1 = WANT_R ? 4 : 0;
    static constexpr bool kReduce = true;
    static constexpr bool kLateIn1 = true;
    static constexpr int kParkWords = 18, kParkCap = SO3_PARK_CAP2;                   // M and Rtrue: 20 KB of LDS per workgroup
    double *loss_sum = nullptr;
    float inv_b = 0.f;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpFrobHead> &rows, RowCtx<NPL> &ctx) const {
        typedef Tr<T> R;
        const T (&m)[9] = rows.a;
        const T (&t)[9] = rows.b;
        T (&dm)[9] = rows.o0;
        T (&r)[9] = rows.o1;
        HardRows<T> h;
        h.hard = quat_rotation<T>(m, r, WANT_DM ? &h.prescale : nullptr);
        h.any = false;
        int base = -1;
        if (__builtin_expect(R::wave_any(h.hard), 0)) {
            base = park_hard_rows<T, NPL, kParkCap>(ctx, h.hard);
            if (base >= 0) {
                park_words<T, NPL, kParkCap, kParkWords, true, 9>(ctx, base, h.hard, 0, m);
            } else {
                h.any = true;
                h.frames = signed_svd<WANT_DM, T>(m);
                T rj[9];
                rotation_from(h.frames, rj);
#pragma unroll
                for (int j = 0; j < 9; ++j) r[j] = R::sel(h.hard, rj[j], r[j]);
            }
        }
        late_in1<T, OpFrobHead, NPL>(ctx, rows.b);                // Rtrue: only now
        if (__builtin_expect(base >= 0, 0)) park_words<T, NPL, kParkCap, kParkWords, false, 9>(ctx, base, h.hard, 9, t);
        T g[9];
        T n2 = R::splat(0.f);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            g[i] = r[i] - t[i];                           // d||Rtrue - R||/dR = (R - Rtrue)/||.||
            n2 = R::fma(g[i], g[i], n2);
        }
        const T inv = R::rsq(R::max(n2, R::splat(1e-37f)));
        const T nrm = n2 * inv;
#pragma unroll
        for (int k = 0; k < NPL; ++k)                     // (a parked row's loss is added when it is redone)
            if (ctx.exists[k] && !(base >= 0 && R::lane_of(h.hard, k))) ctx.acc += static_cast<double>(R::get(nrm, k));
        if (WANT_DM) {
            const T gs = R::sel(R::gt(n2, R::splat(0.f)), inv * R::splat(inv_b), R::splat(0.f));   // zero difference -> zero gradient
#pragma unroll
            for (int i = 0; i < 9; ++i) g[i] = g[i] * gs;
            backward_given_rotation<T>(m, r, g, h, dm);
        }
    }
    template <int NPL>
    __device__ __forceinline__ void redo_parked(RowCtx<NPL> &ctx, int e, bool valid) const {
#pragma clang fp contract(on)         // as the Jacobi path it continues (so3_device.h)
        float m[9], t[9], r[9], g[9];
        parked_words<kParkCap, 9>(ctx.park, e, 0, m);
        parked_words<kParkCap, 9>(ctx.park, e, 9, t);
        const long long row = parked_row<kParkCap, kParkWords>(ctx.park, e);
        const SignedSvd<float> f = signed_svd<WANT_DM, float>(m);
        rotation_from(f, r);
        float n2 = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) { g[i] = r[i] - t[i]; n2 = fmaf(g[i], g[i], n2); }
        const float inv = hw::rsq(fmaxf(n2, 1e-37f));
        if (valid) ctx.acc += static_cast<double>(n2 * inv);
        if (WANT_R && valid) overwrite_row<4, 9>(out1, row, r);
        if (WANT_DM) {
            const float gs = n2 > 0.f ? inv * inv_b : 0.f;
            float dm[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) g[i] = g[i] * gs;
            project_backward(f, g, dm);
            if (valid) overwrite_row<M_BYTES, 9>(out0, row, dm);
        }
    }
    float *loss_mean = nullptr;
    double inv_b_f64 = 0.0;
    __device__ __forceinline__ void finish(double total, bool) const { atomicAdd(loss_sum, total); }
    __device__ __forceinline__ void finish_total(double total, bool) const {
        *loss_sum = total;
        if (loss_mean != nullptr) *loss_mean = static_cast<float>(total * inv_b_f64);
    }
};

// K3': stand-alone Frobenius loss (3D-Pose/loss.py:7-11).  out0 = d(mean loss)/dRpred (optional).
template <bool WANT_GRAD>
struct OpFrobLoss : OpBase {
    static constexpr int kIn0 = 4, kIn1 = 4, kOut0 = WANT_GRAD ? 4 : 0, kOut1 = 0;
    static constexpr bool kReduce = true;
    double *loss_sum = nullptr;
    float inv_b = 0.f;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpFrobLoss> &rows, RowCtx<NPL> &ctx) const {
        typedef Tr<T> R;
        const T (&p)[9] = rows.a;
        const T (&t)[9] = rows.b;
        T (&g)[9] = rows.o0;
        T n2 = R::splat(0.f);
#pragma unroll
        for (int i = 0; i < 9; ++i) { g[i] = p[i] - t[i]; n2 = R::fma(g[i], g[i], n2); }
        const T inv = R::rsq(R::max(n2, R::splat(1e-37f)));
        const T nrm = n2 * inv;
#pragma unroll
        for (int k = 0; k < NPL; ++k)
            if (ctx.exists[k]) ctx.acc += static_cast<double>(R::get(nrm, k));
        const T gs = R::sel(R::gt(n2, R::splat(0.f)), inv * R::splat(inv_b), R::splat(0.f));
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = g[i] * gs;
    }
    float *loss_mean = nullptr;
    double inv_b_f64 = 0.0;
    __device__ __forceinline__ void finish(double total, bool) const { atomicAdd(loss_sum, total); }
    __device__ __forceinline__ void finish_total(double total, bool) const {
        *loss_sum = total;
        if (loss_mean != nullptr) *loss_mean = static_cast<float>(total * inv_b_f64);
    }
};

// acos in float64 to 1.4e-14 rad (the metric is compared at 1e-9 degrees): |c| <= 1/2: pi/2 - asin(c); otherwise through
// asin(sqrt((1 - |c|)/2)).  asin(x) = x + x z g(z), z = x^2 <= 1/4, g a degree-9 polynomial (Chebyshev fit).  A third of the
// device library's acos in instructions: the float64 work of K4 and of the fused K1+K4 is what holds their clocks down.
__device__ __forceinline__ double acos_f64(double c) {
    const double a = __builtin_fabs(c);
    const bool small = a <= 0.5;
    const double z = small ? c * c : (1.0 - a) * 0.5;
    const double x = small ? c : __builtin_sqrt(z);
    double g = 2.80174951579700952e-02;
    g = __builtin_fma(g, z, -3.06358547004551354e-03);
    g = __builtin_fma(g, z, 1.57334373966823808e-02);
    g = __builtin_fma(g, z, 1.31733845957499804e-02);
    g = __builtin_fma(g, z, 1.74436241846820592e-02);
    g = __builtin_fma(g, z, 2.23658829849630453e-02);
    g = __builtin_fma(g, z, 3.03821915977370988e-02);
    g = __builtin_fma(g, z, 4.46428522253018087e-02);
    g = __builtin_fma(g, z, 7.50000000378114595e-02);
    g = __builtin_fma(g, z, 1.66666666666618946e-01);
    const double r = __builtin_fma(x * z, g, x);           // asin(x)
    const double big = c > 0.0 ? r + r : __builtin_fma(-2.0, r, 3.14159265358979323846);
    return small ? 1.57079632679489661923 - r : big;       // NaN in -> NaN out (comparisons false, arithmetic propagates)
}

// K4: theta = acos(clamp((tr(R1^T R2) - 1)/2)) in float64 on float32 data (rotation_representation.py:230-242).
template <bool WANT_DEG, bool WANT_SUM>
struct OpAngle : OpBase {
    static constexpr int kIn0 = 4, kIn1 = 4, kOut0 = 0, kOut1 = 0;
    static constexpr bool kReduce = true;
    double *deg = nullptr, *sum_count = nullptr;
    int32_t *range_flag = nullptr;
    double unit_scale = 1.0;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpAngle> &rows, RowCtx<NPL> &ctx) const {
        const T (&a)[9] = rows.a;
        const T (&b)[9] = rows.b;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            double tr = 0.0;                                 // tr(R1^T R2) = sum_ij R1_ij R2_ij, float64
#pragma unroll
            for (int i = 0; i < 9; ++i) tr = fma(static_cast<double>(Tr<T>::get(a[i], k)), static_cast<double>(Tr<T>::get(b[i], k)), tr);
            const double c_raw = (tr - 1.0) * 0.5;
            if (ctx.exists[k]) ctx.flag |= (c_raw < -1.1 || c_raw > 1.1);   // NaN compares false, as torch.any(...) does
            double c = fmin(fmax(c_raw, -1.0), 1.0);         // torch.clamp ...
            if (c_raw != c_raw) c = c_raw;                   // ... which keeps NaN (fmin/fmax drop it)
            const double ang = acos_f64(c) * unit_scale;
            if (WANT_DEG) {
                const u32x2 bits = __builtin_bit_cast(u32x2, ang);
                __builtin_amdgcn_raw_buffer_store_b64(bits, row_rsrc<8>(deg, ctx.exists[k] ? ctx.unit[k] : 0, ctx.exists[k]), ctx.lane * 8, 0, 0);
            }
            if (WANT_SUM && ctx.exists[k]) ctx.acc += ang;
        }
    }
    double count = 0.0;
    bool store_count = false;      // accumulators pre-zeroed by the caller (so3_*_acc): nobody else writes the row count
    __device__ __forceinline__ void finish(double total, bool any_flag) const {
        if (WANT_SUM) atomicAdd(sum_count, total);
        if (WANT_SUM && store_count && blockIdx.x == 0) sum_count[1] = count;
        if (any_flag && range_flag != nullptr) atomicOr(range_flag, 1);
    }
    __device__ __forceinline__ void finish_total(double total, bool any_flag) const {
        if (WANT_SUM) { sum_count[0] = total; sum_count[1] = count; }
        if (range_flag != nullptr) *range_flag = any_flag ? 1 : 0;
    }
};

// The REDUCED forms of the metric -- sum_b acos(clamp((tr(R1_b^T R2_b) - 1)/2)), what `angle_error(...).mean()` (3D-Pose/main.py:62) and
// the (sum, count) pair of the multi-GPU layer need -- without float64 arithmetic on every row.  The reference casts both
// rotations to float64 before the product (rotation_representation.py:232-233); reproducing that costs 18 v_cvt_f64_f32, 9 float64
// FMAs and a 35-instruction float64 acos per row, 560 cycles per pair of rows on top of K1's 1 750 (docs/history/profiles/r03_valu_rates_f64.txt),
// for a sum whose INPUTS are float32 rotations carrying 1e-7 of orthonormality defect.  Here the same formula runs in packed
// float32 -- trace, cosine, acos as pi/2 - asin(c) for |c| <= 1/2 and through asin(sqrt((1 - |c|)/2)) beyond, a degree-5 polynomial
// (1.2e-9 rad, fitted for float32 evaluation: mean error -3e-10 rad) -- and only the asin part r travels to the float64
// accumulator per row (one v_cvt, one v_add_f64); the multiples of pi/2 are counted on the scalar unit from the lane masks
// (RowCtx::n_half, n_pi) and added once per wave, so no float32 constant's rounding enters the sum.
// Where the cosine is within 5e-7 of +-1 (angles within 1e-3 rad = 0.057 degrees of 0 or 180: acos amplifies the float32 trace's
// 1e-7 of round-off beyond 1e-4 rad there) the row takes the reference's float64 arithmetic as before, under a wave-uniform
// branch (Haar-distributed pairs: one round of 128 rows in twelve) and per row: a row's contribution does not depend on its
// wave-mates.  Outside that band a row's angle differs from the float64 one by at most 2e-7 / sin(theta) rad, without bias
// (round-to-nearest), 3e-8 degrees in the mean of 1M Haar pairs.  The range test (cos outside [-1.1, 1.1]) runs on the float64 cosine, inside the band.
// `skip`: rows that do not count here (the fast path's hard rows: their angle comes from the redo pass).
template <class T, int NPL>
__device__ __forceinline__ void angle_sum_f32(const T (&a)[9], const T (&b)[9], typename Tr<T>::mask skip, RowCtx<NPL> &ctx) {
    typedef Tr<T> R;
    T tr = a[0] * b[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) tr = R::fma(a[i], b[i], tr);
    const T c = (tr - R::splat(1.f)) * R::splat(0.5f);
    const T ac = R::abs(c);
    // every comparison is false for NaN: a NaN cosine takes the sqrt branch and stays NaN (torch.clamp keeps NaN, acos returns it)
    const typename R::mask small = R::le(ac, R::splat(0.5f));
    const T zb = R::sel(R::gt(ac, R::splat(1.f)), R::splat(0.f), R::fma(ac, R::splat(-0.5f), R::splat(0.5f)));    // clamp: |c| > 1 -> angle 0 or pi
    const T z = R::sel(small, c * c, zb);
    const T x = R::sel(small, c, R::sqrt(zb));
    T g = R::fma(z, R::splat(3.392098099e-02f), R::splat(1.700584404e-02f));
    g = R::fma(g, z, R::splat(3.113190830e-02f));
    g = R::fma(g, z, R::splat(4.459662735e-02f));
    g = R::fma(g, z, R::splat(7.500103116e-02f));
    g = R::fma(g, z, R::splat(1.666666567e-01f));
    const T r = R::fma(x * z, g, x);                                   // asin(x)
    // theta = pi/2 - r (small) | 2 r (c > 1/2) | pi - 2 r (c < -1/2): r times -1 / 2 / -2 here, the constants by count
    const typename R::mask neg = R::gt(R::splat(0.f), c);
    const T rm = r * R::sel(small, R::splat(-1.f), R::sel(neg, R::splat(-2.f), R::splat(2.f)));
    const typename R::mask band = R::gt(ac, R::splat(0.9999995f));     // cosine within 5e-7 of +-1 (or beyond): float64 for this row
    bool any_band = false;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        if (!ctx.exists[k]) continue;                                   // wave-uniform: the phantom unit of an odd tail
        const bool counts = !R::lane_of(skip, k);
        ctx.acc += counts ? static_cast<double>(R::get(rm, k)) : 0.0;
        ctx.n_half += __builtin_popcountll(__builtin_amdgcn_ballot_w64(counts && R::lane_of(small, k)));
        ctx.n_pi += __builtin_popcountll(__builtin_amdgcn_ballot_w64(counts && !R::lane_of(small, k) && R::lane_of(neg, k)));
        any_band |= wave_any(counts && R::lane_of(band, k));
    }
    if (__builtin_expect(any_band, 0)) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            double t64 = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i) t64 = fma(static_cast<double>(R::get(a[i], k)), static_cast<double>(R::get(b[i], k)), t64);
            const double c_raw = (t64 - 1.0) * 0.5;
            // the reference's range test (rotation_representation.py:236-239), on the float64 cosine like the reference's: every row it can
            // fire for lies inside the band (round 4 tested the float32 cosine: a row at 1.1 +- 1e-7 could raise or not raise differently)
            if (ctx.exists[k] && R::lane_of(band, k) && !R::lane_of(skip, k)) ctx.flag |= (c_raw < -1.1 || c_raw > 1.1);
            const double c64 = fmin(fmax(c_raw, -1.0), 1.0);                   // (a band row's cosine is finite)
            // what the row has contributed above is K + rm with K = 0 or pi: replace it by the float64 angle
            const double base = R::lane_of(neg, k) ? 3.14159265358979323846 : 0.0;
            const double corr = acos_f64(c64) - base - static_cast<double>(R::get(rm, k));
            if (ctx.exists[k] && R::lane_of(band, k) && !R::lane_of(skip, k)) ctx.acc += corr;
        }
    }
}

// K1 + K4 fused: theta_b = angle(proj(M_b), T_b) without materialising R (72 B read per row, nothing written
// unless the per-row angles or R are requested).  The evaluation step of the reference,
// `angle_error(func[rot_rep](out), R).mean()` (3D-Pose/main.py:60-62,110-112), in one launch.
// F32SUM (only without per-row angles): the sum through angle_sum_f32 above; otherwise every row in the reference's float64.
template <int M_BYTES, bool WANT_R, bool WANT_DEG, bool WANT_SUM, bool F32SUM = false>
struct OpProjectAngle : OpBase {
    static_assert(!F32SUM || (!WANT_DEG && WANT_SUM), "the float32 angle sum is a reduced form");
    static constexpr int kIn0 = M_BYTES, kIn1 = 4, kOut0 = WANT_R ? 4 : 0, kOut1 = 0;
    static constexpr bool kReduce = true;
    static constexpr bool kLateIn1 = true;
    static constexpr int kParkWords = 18, kParkCap = SO3_PARK_CAP2;                   // M and Rtrue: 20 KB of LDS per workgroup
    static constexpr bool kAngleConstants = F32SUM;
    double *deg = nullptr, *sum_count = nullptr;
    int32_t *range_flag = nullptr;
    double unit_scale = 1.0;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpProjectAngle> &rows, RowCtx<NPL> &ctx) const {
        typedef Tr<T> R;
        T r[9];
        const typename R::mask hard = quat_rotation<T>(rows.a, r);
        int base = -1;
        if (__builtin_expect(R::wave_any(hard), 0)) {
            base = park_hard_rows<T, NPL, kParkCap>(ctx, hard);
            if (base >= 0) {
                park_words<T, NPL, kParkCap, kParkWords, true, 9>(ctx, base, hard, 0, rows.a);
            } else {
                T rj[9];
                rotation_from(signed_svd<false, T>(rows.a), rj);
#pragma unroll
                for (int j = 0; j < 9; ++j) r[j] = R::sel(hard, rj[j], r[j]);
            }
        }
        late_in1<T, OpProjectAngle, NPL>(ctx, rows.b);            // Rtrue: only now
        if (__builtin_expect(base >= 0, 0)) park_words<T, NPL, kParkCap, kParkWords, false, 9>(ctx, base, hard, 9, rows.b);
        // rows that do not count here: the parked ones (their rotation is not in r; redo_parked answers for them)
        const typename R::mask skip = base >= 0 ? hard : R::gt(R::splat(0.f), R::splat(1.f));
        if (WANT_R) {
#pragma unroll
            for (int i = 0; i < 9; ++i) rows.o0[i] = r[i];
        }
        if constexpr (F32SUM) {
            angle_sum_f32<T, NPL>(r, rows.b, skip, ctx);     // radians; scale_partial() turns the workgroup's total into the unit asked for
            return;
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            double tr = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i) tr = fma(static_cast<double>(Tr<T>::get(r[i], k)), static_cast<double>(Tr<T>::get(rows.b[i], k)), tr);
            const double c_raw = (tr - 1.0) * 0.5;
            const bool counts = ctx.exists[k] && !R::lane_of(skip, k);
            if (counts) ctx.flag |= (c_raw < -1.1 || c_raw > 1.1);
            double c = fmin(fmax(c_raw, -1.0), 1.0);
            if (c_raw != c_raw) c = c_raw;
            const double ang = acos_f64(c) * unit_scale;
            if (WANT_DEG) {
                const u32x2 bits = __builtin_bit_cast(u32x2, ang);
                __builtin_amdgcn_raw_buffer_store_b64(bits, row_rsrc<8>(deg, ctx.exists[k] ? ctx.unit[k] : 0, ctx.exists[k]), ctx.lane * 8, 0, 0);
            }
            if (WANT_SUM && counts) ctx.acc += ang;
        }
    }
    template <int NPL>
    __device__ __forceinline__ void redo_parked(RowCtx<NPL> &ctx, int e, bool valid) const {
#pragma clang fp contract(on)         // as the Jacobi path it continues (so3_device.h)
        float m[9], t[9], r[9];
        parked_words<kParkCap, 9>(ctx.park, e, 0, m);
        parked_words<kParkCap, 9>(ctx.park, e, 9, t);
        const long long row = parked_row<kParkCap, kParkWords>(ctx.park, e);
        rotation_from(signed_svd<false, float>(m), r);
        if (WANT_R && valid) overwrite_row<4, 9>(out0, row, r);
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) tr = fma(static_cast<double>(r[i]), static_cast<double>(t[i]), tr);
        const double c_raw = (tr - 1.0) * 0.5;
        if (valid) ctx.flag |= (c_raw < -1.1 || c_raw > 1.1);
        double c = fmin(fmax(c_raw, -1.0), 1.0);
        if (c_raw != c_raw) c = c_raw;
        const double rad = acos_f64(c);
        if (WANT_DEG && valid) __builtin_nontemporal_store(rad * unit_scale, deg + row);
        if (WANT_SUM && valid) ctx.acc += F32SUM ? rad : rad * unit_scale;          // (the float32 sum is in radians until scale_partial)
    }
    double count = 0.0;
    bool store_count = false;      // accumulators pre-zeroed by the caller (so3_*_acc): nobody else writes the row count
    __device__ __forceinline__ double scale_partial(double t) const { return F32SUM ? t * unit_scale : t; }   // angle_sum_f32 sums radians
    __device__ __forceinline__ void finish(double total, bool any_flag) const {
        if (WANT_SUM) atomicAdd(sum_count, total);
        if (WANT_SUM && store_count && blockIdx.x == 0) sum_count[1] = count;
        if (any_flag && range_flag != nullptr) atomicOr(range_flag, 1);
    }
    __device__ __forceinline__ void finish_total(double total, bool any_flag) const {
        if (WANT_SUM) { sum_count[0] = total; sum_count[1] = count; }
        if (range_flag != nullptr) *range_flag = any_flag ? 1 : 0;
    }
};

// K4': float32 radians, tr(m1 m2^T), clamp to [lo, hi], acos: compute_geodesic_distance_from_two_matrices (rotation_representation.py:209-227:
// lo, hi = -1, 1) and geodesic(R1, R2, reduction) (point_cloud/main.py:61-73: -1 + 1e-7, 1 - 1e-7, and the sum over the batch).
// SUM: the angles are also summed (float64 per lane, one atomic per workgroup onto `sum`); theta may then be null.
template <bool SUM>
struct OpGeodesic : OpBase {
    static constexpr int kIn0 = 4, kIn1 = 4, kOut0 = 0, kOut1 = 0;
    static constexpr bool kReduce = SUM;
    float *theta = nullptr;
    double *sum = nullptr;
    float lo = -1.f, hi = 1.f;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpGeodesic> &rows, RowCtx<NPL> &ctx) const {
        typedef Tr<T> R;
        const T (&a)[9] = rows.a;
        const T (&b)[9] = rows.b;
        // diagonal of m1 m2^T, summed in the reference's order: m00 + m11 + m22
        const T d0 = R::fma(a[2], b[2], R::fma(a[1], b[1], a[0] * b[0]));
        const T d1 = R::fma(a[5], b[5], R::fma(a[4], b[4], a[3] * b[3]));
        const T d2 = R::fma(a[8], b[8], R::fma(a[7], b[7], a[6] * b[6]));
        const T cs = (d0 + d1 + d2 - R::splat(1.f)) * R::splat(0.5f);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            float c = R::get(cs, k);
            c = (c > hi) ? hi : c;       // torch.min / torch.max with a constant, torch.clamp: NaN stays NaN
            c = (c < lo) ? lo : c;
            const float th = acosf(c);
            if (!SUM || theta != nullptr)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(th), row_rsrc<4>(theta, ctx.exists[k] ? ctx.unit[k] : 0, ctx.exists[k]),
                                                      ctx.lane * 4, 0, 0);
            if (SUM && ctx.exists[k]) ctx.acc += static_cast<double>(th);
        }
    }
    float *result = nullptr;     // with a workspace: the reduced value, float32 like the reference's tensor
    double scale = 1.0;          // 1 / B for "mean"
    __device__ __forceinline__ void finish(double total, bool) const { if (SUM) atomicAdd(sum, total); }
    __device__ __forceinline__ void finish_total(double total, bool) const {
        *sum = total;
        if (result != nullptr) *result = static_cast<float>(total * scale);
    }
};

// K4b: the gradient of the three metric spellings -- what autograd computes through geodesic(R1, R2, reduction)
// (point_cloud/main.py:61-73, whose eps = 1e-7 exists for exactly this gradient: the comment at :64), through
// compute_geodesic_distance_from_two_matrices (rotation_representation.py:209-227) and through angle_error (:230-242) when a
// training loop uses one of them as its `lossfunc` (point_cloud/main.py:194-197, UPNA/main.py:56-59).  With
//     c_b = (sum_ij R1_b,ij R2_b,ij - 1) / 2,      theta_b = unit * acos(clamp(c_b, lo, hi)),
// tr(R1 R2^T) = tr(R1^T R2) is symmetric in the two arguments:
//     dR1_b = h_b R2_b,  dR2_b = h_b R1_b,   h_b = g_b * unit * (-1 / sqrt(1 - c_b^2)) / 2   where lo <= c_b <= hi,   0 outside
// (torch.clamp's and torch.min / max's backward fill the gradient with 0 outside the clamp, so the reference never multiplies 0 by
// the infinite slope of acos there; on the boundary c_b = +-1 EXACTLY the reference returns -+inf and this kernel 0).  g_b is the
// upstream gradient: per row (GRAD 1: float32, GRAD 2: float64 as two dwords per row -- the engine's images are dwords) or ONE
// value in device memory that every row shares (GRAD 0: the 0-dim tensor autograd hands to a "mean" / "sum"), divided by `div`
// (the mean's B; torch divides, it does not multiply by 1/B).
// F64MATH = angle_error's spelling: both rotations cast to float64 before the product (:232-233), every step in float64, the
// result rounded to float32 ONCE (the backward of `.double()`); otherwise float32 throughout in the order of the float32 graph:
// trace as OpGeodesic's, -c c + 1, rsqrt, times the upstream gradient, the clamp's mask, / 2, times the other rotation.
// in0 = R1, in1 = R2, in2 = g (GRAD != 0); out0 = dR1, out1 = dR2 (BOTH).  One gradient alone: the host swaps R1 and R2.
template <int GRAD, bool F64MATH, bool BOTH>
struct OpAngleBwd : OpBase {
    static_assert(GRAD == 0 || GRAD == (F64MATH ? 2 : 1), "the upstream gradient has the dtype of the metric's result");
    static constexpr int kIn0 = 4, kIn1 = 4, kIn2 = GRAD != 0 ? 4 : 0, kOut0 = 4, kOut1 = BOTH ? 4 : 0;
    static constexpr int kIn2N = GRAD == 2 ? 2 : 1;
    const void *gscalar = nullptr;       // GRAD 0: one float32 (F64MATH: float64) in device memory
    double lo = -1.0, hi = 1.0, unit = 1.0, div = 1.0;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpAngleBwd> &rows, RowCtx<NPL> &) const {
#pragma clang fp contract(off)            // the float32 graph rounds c c and 1 - c c separately
        typedef Tr<T> R;
        const T (&a)[9] = rows.a;
        const T (&b)[9] = rows.b;
        if constexpr (!F64MATH) {
            const T d0 = R::fma(a[2], b[2], R::fma(a[1], b[1], a[0] * b[0]));       // the diagonal of m1 m2^T, as OpGeodesic sums it
            const T d1 = R::fma(a[5], b[5], R::fma(a[4], b[4], a[3] * b[3]));
            const T d2 = R::fma(a[8], b[8], R::fma(a[7], b[7], a[6] * b[6]));
            const T cs = (d0 + d1 + d2 - R::splat(1.f)) * R::splat(0.5f);
            const float lof = static_cast<float>(lo), hif = static_cast<float>(hi), divf = static_cast<float>(div), unitf = static_cast<float>(unit);
            T h = R::splat(0.f);
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const float c = R::get(cs, k);
                float g;
                if constexpr (GRAD == 0) g = *static_cast<const float *>(gscalar); else g = R::get(rows.c[0], k);
                g = g / divf;
                const float om = 1.f - c * c;
                float s = (c >= lof && c <= hif && om > 0.f) ? g * -__builtin_amdgcn_rsqf(om) : 0.f;
                if (c != c) s = c;                                   // NaN in -> NaN out
                R::set(h, k, s * unitf * 0.5f);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                rows.o0[i] = h * b[i];
                if constexpr (BOTH) rows.o1[i] = h * a[i];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                double tr = 0.0;                                     // tr(R1^T R2) in float64, as OpAngle
#pragma unroll
                for (int i = 0; i < 9; ++i) tr = fma(static_cast<double>(R::get(a[i], k)), static_cast<double>(R::get(b[i], k)), tr);
                const double c = (tr - 1.0) * 0.5;
                double g;
                if constexpr (GRAD == 0) g = *static_cast<const double *>(gscalar);
                else g = __hiloint2double(__float_as_int(R::get(rows.c[1], k)), __float_as_int(R::get(rows.c[0], k)));
                g = g / div;
                const double om = 1.0 - c * c;
                double h = (c >= lo && c <= hi && om > 0.0) ? (g * unit) * (-1.0 / __builtin_sqrt(om)) * 0.5 : 0.0;
                if (c != c) h = c;
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    R::set(rows.o0[i], k, static_cast<float>(h * static_cast<double>(R::get(b[i], k))));
                    if constexpr (BOTH) R::set(rows.o1[i], k, static_cast<float>(h * static_cast<double>(R::get(a[i], k))));
                }
            }
        }
    }
};

#endif  // !SO3_HOST_MODEL

// ---- next row f2: the 6D Gram-Schmidt head (rotation_representation.py:21-36) --------------------------------
// x = a/|a|,  z = (x x b)/|x x b|,  y = z x x,  R = [x y z] (columns); a, b = the two halves of the 6-vector.
template <class T> __device__ __forceinline__ void ortho6d_forward(V3<T> a, V3<T> b, T (&r)[9]) {
    typedef Tr<T> R;
    const V3<T> x = scale<T>(a, R::rsq(dot(a, a)));
    const V3<T> w = cross<T>(x, b);
    const V3<T> z = scale<T>(w, R::rsq(dot(w, w)));
    const V3<T> y = cross<T>(z, x);
    r[0] = x.x; r[1] = y.x; r[2] = z.x;
    r[3] = x.y; r[4] = y.y; r[5] = z.y;
    r[6] = x.z; r[7] = y.z; r[8] = z.z;
}
// G = dL/dR (row-major)  ->  dL/da, dL/db
template <class T> __device__ __forceinline__ void ortho6d_backward(V3<T> a, V3<T> b, const T (&g)[9], V3<T> &ga, V3<T> &gb) {
    typedef Tr<T> R;
    const T ia = R::rsq(dot(a, a));
    const V3<T> x = scale<T>(a, ia);
    const V3<T> w = cross<T>(x, b);
    const T iw = R::rsq(dot(w, w));
    const V3<T> z = scale<T>(w, iw);
    const V3<T> gx = mk<T>(g[0], g[3], g[6]), gy = mk<T>(g[1], g[4], g[7]), gz = mk<T>(g[2], g[5], g[8]);   // columns of G
    // y = z x x :  gz += x x gy ,  gx += gy x z
    const V3<T> gzt = axpy<T>(R::splat(1.f), cross<T>(x, gy), gz);
    V3<T> gxt = axpy<T>(R::splat(1.f), cross<T>(gy, z), gx);
    // z = w/|w| :  gw = (gz - z (z.gz)) / |w|
    const V3<T> gw = scale<T>(axpy<T>(-dot(z, gzt), z, gzt), iw);
    // w = x x b :  gx += b x gw ,  gb = gw x x
    gxt = axpy<T>(R::splat(1.f), cross<T>(b, gw), gxt);
    gb = cross<T>(gw, x);
    // x = a/|a| :  ga = (gx - x (x.gx)) / |a|
    ga = scale<T>(axpy<T>(-dot(x, gxt), x, gxt), ia);
}

struct OpOrtho6d : OpBase {
    static constexpr int kIn0 = 4, kIn1 = 0, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 6;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpOrtho6d> &rows, RowCtx<NPL> &) const {
        const T (&p)[6] = rows.a;
        ortho6d_forward<T>(mk<T>(p[0], p[1], p[2]), mk<T>(p[3], p[4], p[5]), rows.o0);
    }
};

// Its backward: in0 = poses (B,6), in1 = G = dL/dR (B,9), out0 = dL/dposes (B,6).
struct OpOrtho6dBwd : OpBase {
    static constexpr int kIn0 = 4, kIn1 = 4, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 6, kOut0N = 6;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpOrtho6dBwd> &rows, RowCtx<NPL> &) const {
        const T (&p)[6] = rows.a;
        T (&dp)[6] = rows.o0;
        V3<T> ga, gb;
        ortho6d_backward<T>(mk<T>(p[0], p[1], p[2]), mk<T>(p[3], p[4], p[5]), rows.b, ga, gb);
        dp[0] = ga.x; dp[1] = ga.y; dp[2] = ga.z;
        dp[3] = gb.x; dp[4] = gb.y; dp[5] = gb.z;
    }
};

// ---- next row f5: the other heads of the reference's dispatch tables (Comparison/models.py:18-19,
// 3D-Pose/main.py:46, rotation_representation.py:323-324), forward and backward ----------------------------------
// Every *Bwd operation: in0 = the head's input (B,N), in1 = G = dL/dR (B,9), out0 = dL/dinput (B,N).

// 'Quat': q = (w,x,y,z) -> n = q / max(|q|, 1e-8) -> R(n)   (rotation_representation.py:39-50, 137-171)
constexpr float kQuatMinNorm = 1e-8f;
template <bool BWD> struct OpQuat : OpBase {
    static constexpr int kIn0 = 4, kIn1 = BWD ? 4 : 0, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 4, kOut0N = BWD ? 4 : 9;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpQuat> &rows, RowCtx<NPL> &) const {
        typedef Tr<T> R;
        const T (&q)[4] = rows.a;
        const T n2 = R::fma(q[3], q[3], R::fma(q[2], q[2], R::fma(q[1], q[1], q[0] * q[0])));
        const T mag = R::max(R::sqrt(n2), R::splat(kQuatMinNorm));
        const T im = R::rcp(mag);
        const T w = q[0] * im, x = q[1] * im, y = q[2] * im, z = q[3] * im;
        const T two = R::splat(2.f);
        if constexpr (!BWD) {
            T (&r)[9] = rows.o0;
            const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, xw = x * w, yw = y * w, zw = z * w;
            const T one = R::splat(1.f);
            r[0] = one - two * (yy + zz); r[1] = two * (xy - zw);       r[2] = two * (xz + yw);
            r[3] = two * (xy + zw);       r[4] = one - two * (xx + zz); r[5] = two * (yz - xw);
            r[6] = two * (xz - yw);       r[7] = two * (yz + xw);       r[8] = one - two * (xx + yy);
        } else {
            const T (&g)[9] = rows.b;
            T (&dq)[4] = rows.o0;
            // dL/dn from the nine polynomial entries
            const T s12 = g[1] + g[3], s02 = g[2] + g[6], s21 = g[5] + g[7];      // symmetric parts
            const T a0 = g[7] - g[5], a1 = g[2] - g[6], a2 = g[3] - g[1];          // antisymmetric parts
            const T gw = two * R::fma(x, a0, R::fma(y, a1, z * a2));
            const T gx = two * R::fma(w, a0, R::fma(y, s12, R::fma(z, s02, -two * x * (g[4] + g[8]))));
            const T gy = two * R::fma(w, a1, R::fma(x, s12, R::fma(z, s21, -two * y * (g[0] + g[8]))));
            const T gz = two * R::fma(w, a2, R::fma(x, s02, R::fma(y, s21, -two * z * (g[0] + g[4]))));
            // n = q/|q| (or q/1e-8 when clamped: the divisor is then a constant)
            const typename R::mask clamped = R::le(R::sqrt(n2), R::splat(kQuatMinNorm));
            const T proj = R::sel(clamped, R::splat(0.f), R::fma(w, gw, R::fma(x, gx, R::fma(y, gy, z * gz))));
            dq[0] = R::fma(-proj, w, gw) * im;
            dq[1] = R::fma(-proj, x, gx) * im;
            dq[2] = R::fma(-proj, y, gy) * im;
            dq[3] = R::fma(-proj, z, gz) * im;
        }
    }
};

// 'Euler': angles (e0, e1, e2) -> R, with c1,s1 of e0; c2,s2 of e2; c3,s3 of e1   (rotation_representation.py:92-113)
template <bool BWD> struct OpEuler : OpBase {
    static constexpr int kIn0 = 4, kIn1 = BWD ? 4 : 0, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 3, kOut0N = BWD ? 3 : 9;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpEuler> &rows, RowCtx<NPL> &) const {
        typedef Tr<T> R;
        const T (&e)[3] = rows.a;
        T c1, s1, c2, s2, c3, s3;
        R::sincos(e[0], s1, c1);
        R::sincos(e[2], s2, c2);
        R::sincos(e[1], s3, c3);
        T r[9];
        r[0] = c2 * c3;                     r[1] = -s2;     r[2] = c2 * s3;
        r[3] = R::fma(c1 * s2, c3, s1 * s3); r[4] = c1 * c2; r[5] = R::fma(c1 * s2, s3, -(s1 * c3));
        r[6] = R::fma(s1 * s2, c3, -(c1 * s3)); r[7] = s1 * c2; r[8] = R::fma(s1 * s2, s3, c1 * c3);
        if constexpr (!BWD) {
#pragma unroll
            for (int i = 0; i < 9; ++i) rows.o0[i] = r[i];
        } else {
            const T (&g)[9] = rows.b;
            T (&de)[3] = rows.o0;
            // d/de0: row2' = -row3, row3' = row2.   d/de1: column0' = -column2, column2' = column0.
            de[0] = R::fma(g[8], r[5], R::fma(g[7], r[4], g[6] * r[3])) - R::fma(g[5], r[8], R::fma(g[4], r[7], g[3] * r[6]));
            de[1] = R::fma(g[8], r[6], R::fma(g[5], r[3], g[2] * r[0])) - R::fma(g[6], r[8], R::fma(g[3], r[5], g[0] * r[2]));
            // d/de2 (c2' = -s2, s2' = c2)
            const T d0 = -(s2 * c3), d1 = -c2, d2 = -(s2 * s3);
            const T d3 = c1 * c2 * c3, d4 = -(c1 * s2), d5 = c1 * c2 * s3;
            const T d6 = s1 * c2 * c3, d7 = -(s1 * s2), d8 = s1 * c2 * s3;
            de[2] = R::fma(g[0], d0, R::fma(g[1], d1, R::fma(g[2], d2, R::fma(g[3], d3, R::fma(g[4], d4,
                    R::fma(g[5], d5, R::fma(g[6], d6, R::fma(g[7], d7, g[8] * d8))))))));
        }
    }
};

// '5D': a -> 6D through the stereographic un-projection of a[2:5] * (1+sqrt2, 1+sqrt2, sqrt2), then the 6D head
// (rotation_representation.py:69-90, 118-134).  With v = a[2:5] * scale, s = |v|^2 the reference's normalised
// 4-vector u/|u[1:]| is ((s-1)/(2|v|), v/|v|):  x_raw = (a0, a1, (s-1)/(2|v|)),  y_raw = v/|v|.
template <bool BWD> struct OpOrtho5d : OpBase {
    static constexpr int kIn0 = 4, kIn1 = BWD ? 4 : 0, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 5, kOut0N = BWD ? 5 : 9;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpOrtho5d> &rows, RowCtx<NPL> &) const {
        typedef Tr<T> R;
        const T (&a)[5] = rows.a;
        const T k0 = R::splat(2.41421356237309505f), k2 = R::splat(1.41421356237309505f);
        const V3<T> v = mk<T>(a[2] * k0, a[3] * k0, a[4] * k2);
        const T s = dot(v, v);
        const T ir = R::rsq(s);
        const V3<T> vh = scale<T>(v, ir);
        const V3<T> xr = mk<T>(a[0], a[1], (s - R::splat(1.f)) * R::splat(0.5f) * ir);
        if constexpr (!BWD) {
            ortho6d_forward<T>(xr, vh, rows.o0);
        } else {
            T (&da)[5] = rows.o0;
            V3<T> gx, gy;
            ortho6d_backward<T>(xr, vh, rows.b, gx, gy);
            // d x_raw.z / dv = v (s+1) / (2 s |v|) ;  y_raw = v/|v|
            const T kx = gx.z * (s + R::splat(1.f)) * R::splat(0.5f) * ir * ir * ir;
            const V3<T> gv = axpy<T>(kx, v, scale<T>(axpy<T>(-dot(vh, gy), vh, gy), ir));
            da[0] = gx.x; da[1] = gx.y;
            da[2] = gv.x * k0; da[3] = gv.y * k0; da[4] = gv.z * k2;
        }
    }
};

// '3D': the so(3) exponential map with PyTorch3D's clamp, theta = sqrt(max(|v|^2, 1e-4))
// (rotation_representation.py:245-275, 278-321):  R = I + (sin t / t) K + ((1 - cos t) / t^2) K^2,  K = hat(v),
// K^2 = v v^T - |v|^2 I  (the un-clamped |v|^2, as the reference's bmm gives).
constexpr float kExpMapEps = 1e-4f;
template <bool BWD> struct OpExpMap : OpBase {
    static constexpr int kIn0 = 4, kIn1 = BWD ? 4 : 0, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 3, kOut0N = BWD ? 3 : 9;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpExpMap> &rows, RowCtx<NPL> &) const {
        typedef Tr<T> R;
        const T (&p)[3] = rows.a;
        const V3<T> v = mk<T>(p[0], p[1], p[2]);
        const T nrm = dot(v, v);
        const T t2 = R::max(nrm, R::splat(kExpMapEps));
        const T it = R::rsq(t2);
        const T t = t2 * it;
        T st, ct;
        R::sincos(t, st, ct);
        const T sh = R::sin(t * R::splat(0.5f));
        const T f1 = st * it;
        const T f2 = R::splat(2.f) * sh * sh * it * it;              // (1 - cos t)/t^2 without the cancellation
        if constexpr (!BWD) {
            T (&r)[9] = rows.o0;
            const T one = R::splat(1.f);
            const T xy = f2 * v.x * v.y, xz = f2 * v.x * v.z, yz = f2 * v.y * v.z;
            r[0] = R::fma(f2, R::fma(v.x, v.x, -nrm), one); r[1] = R::fma(-f1, v.z, xy); r[2] = R::fma(f1, v.y, xz);
            r[3] = R::fma(f1, v.z, xy); r[4] = R::fma(f2, R::fma(v.y, v.y, -nrm), one); r[5] = R::fma(-f1, v.x, yz);
            r[6] = R::fma(-f1, v.y, xz); r[7] = R::fma(f1, v.x, yz); r[8] = R::fma(f2, R::fma(v.z, v.z, -nrm), one);
        } else {
            const T (&g)[9] = rows.b;
            T (&dv)[3] = rows.o0;
            const V3<T> a = mk<T>(g[7] - g[5], g[2] - g[6], g[3] - g[1]);                  // <G, dK/dv>
            const T tr = g[0] + g[4] + g[8];
            const V3<T> sv = mk<T>(R::fma(g[1] + g[3], v.y, R::fma(g[2] + g[6], v.z, (g[0] + g[0]) * v.x)),
                                   R::fma(g[1] + g[3], v.x, R::fma(g[5] + g[7], v.z, (g[4] + g[4]) * v.y)),
                                   R::fma(g[2] + g[6], v.x, R::fma(g[5] + g[7], v.y, (g[8] + g[8]) * v.z)));   // (G + G^T) v
            const T df1 = dot(v, a);                                                       // dL/dfac1 = <G, K>
            const T df2 = R::fma(R::splat(0.5f), dot(v, sv), -(nrm * tr));                 // dL/dfac2 = <G, K^2>
            // (dfac/dtheta)/theta: closed forms cancel badly for small theta -> series below theta = 1
            const T it3 = it * it * it;
            const T a1c = R::fma(t, ct, -st) * it3;
            const T a2c = R::fma(t, st, -(R::splat(4.f) * sh * sh)) * it3 * it;
            const T a1s = R::fma(t2, R::fma(t2, R::fma(t2, R::fma(t2, R::splat(-1.f / 3991680.f), R::splat(1.f / 45360.f)),
                                 R::splat(-1.f / 840.f)), R::splat(1.f / 30.f)), R::splat(-1.f / 3.f));
            const T a2s = R::fma(t2, R::fma(t2, R::fma(t2, R::fma(t2, R::splat(-1.f / 47900160.f), R::splat(1.f / 453600.f)),
                                 R::splat(-1.f / 6720.f)), R::splat(1.f / 180.f)), R::splat(-1.f / 12.f));
            const typename R::mask small = R::le(t2, R::splat(1.f));
            const T a1 = R::sel(small, a1s, a1c), a2 = R::sel(small, a2s, a2c);
            const T via_theta = R::sel(R::ge(nrm, R::splat(kExpMapEps)), R::fma(df1, a1, df2 * a2), R::splat(0.f));
            const T m2 = -(tr + tr);
            dv[0] = R::fma(via_theta, v.x, R::fma(f1, a.x, f2 * R::fma(m2, v.x, sv.x)));
            dv[1] = R::fma(via_theta, v.y, R::fma(f1, a.y, f2 * R::fma(m2, v.y, sv.y)));
            dv[2] = R::fma(via_theta, v.z, R::fma(f1, a.z, f2 * R::fma(m2, v.z, sv.z)));
        }
    }
};

// ---- next row f1: the SE(3) update of Iterative/utility.py:90-128 (calculate_T_pred), fused after the head -----
// in0 = network output (B,12): nine numbers for the rotation head, then (vx, vy, vz);  in1 = T_init (B,4,4).
//   dR = proj(out[:, :9]);  R_new = dR R_k;  z_new = vz z_k;  x_new = (vx/fx + x_k/z_k) z_new;  y likewise;
//   T_pred = [[R_new, t_new], [0 0 0 1]]      (what the reference's `combine`, utility.py:63-71, intends).
struct OpSe3Update : OpBase {
    static constexpr int kIn0 = 4, kIn1 = 4, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 12, kIn1N = 16, kOut0N = 16;
    float inv_fx = 0.f, inv_fy = 0.f;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpSe3Update> &rows, RowCtx<NPL> &) const {
        typedef Tr<T> R;
        const T (&o)[12] = rows.a;
        const T (&ti)[16] = rows.b;
        T (&tp)[16] = rows.o0;
        T m[9], dr[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = o[i];
        project_rotation<T>(m, dr);                            // utility.py:105
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)                       // R_new = dR R_k  (:124)
                tp[4 * i + j] = R::fma(dr[3 * i + 2], ti[8 + j], R::fma(dr[3 * i + 1], ti[4 + j], dr[3 * i] * ti[j]));
        const T zk = ti[11], iz = R::rcp(zk);
        const T zn = o[11] * zk;                              // :116
        tp[3] = R::fma(o[9], R::splat(inv_fx), ti[3] * iz) * zn;    // :120
        tp[7] = R::fma(o[10], R::splat(inv_fy), ti[7] * iz) * zn;   // :121
        tp[11] = zn;
        tp[12] = R::splat(0.f); tp[13] = R::splat(0.f); tp[14] = R::splat(0.f);
        tp[15] = R::splat(1.f);
    }
};

// Its backward w.r.t. the network output: in0 = output (B,12), in1 = T_init (B,16), in2 = G = dL/dT_pred (B,16).
struct OpSe3UpdateBwd : OpBase {
    static constexpr int kIn0 = 4, kIn1 = 4, kIn2 = 4, kOut0 = 4, kOut1 = 0;
    static constexpr int kIn0N = 12, kIn1N = 16, kIn2N = 16, kOut0N = 12;
    float inv_fx = 0.f, inv_fy = 0.f;
    template <class T, int NPL>
    __device__ __forceinline__ void compute(Rows<T, OpSe3UpdateBwd> &rows, RowCtx<NPL> &) const {
        typedef Tr<T> R;
        const T (&o)[12] = rows.a;
        const T (&ti)[16] = rows.b;
        const T (&g)[16] = rows.c;
        T (&d)[12] = rows.o0;
        T m[9], gdr[9], dm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = o[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)                       // dL/d(dR) = G_R R_k^T
                gdr[3 * i + j] = R::fma(g[4 * i + 2], ti[4 * j + 2], R::fma(g[4 * i + 1], ti[4 * j + 1], g[4 * i] * ti[4 * j]));
        project_backward_rows<T>(m, gdr, dm);
#pragma unroll
        for (int i = 0; i < 9; ++i) d[i] = dm[i];
        const T zk = ti[11], iz = R::rcp(zk);
        const T zn = o[11] * zk;
        const T ax = R::fma(o[9], R::splat(inv_fx), ti[3] * iz), ay = R::fma(o[10], R::splat(inv_fy), ti[7] * iz);
        d[9] = g[3] * zn * R::splat(inv_fx);
        d[10] = g[7] * zn * R::splat(inv_fy);
        d[11] = zk * R::fma(g[7], ay, R::fma(g[3], ax, g[11]));
    }
};

}  // namespace so3
