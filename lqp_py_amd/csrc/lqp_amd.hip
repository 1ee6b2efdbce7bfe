// C ABI of the MI355X box-QP ADMM layer: workspace carving, launch
// orchestration, status reporting.  See include/lqp_amd.h for the contract.
#include "../../include/lqp_amd.h"
#include "lqp_boxqp.hpp"
#include "lqp_unroll.hpp"
#ifdef LQP_SPLIT_BUILD
// split build (lqp_py_amd/_lib.py build_library, tools/gen_split_build.py): the kernel instances are compiled in the
// translation units csrc/split/lqp_tu_*.hip; here they are only declared
#include "split/lqp_extern.inc"
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>
#include <string>
#include <unordered_map>

using namespace lqp;

namespace {

constexpr int kRing = 2048;          // per-check counter slots (a continuation launch holds kRing / 2 checks: 10 000 iterations at a check every 10 in ONE launch)
// pivoted LU: one panel row per thread to 1024 rows, two to 2048, four (float32, panels of 4 columns) to 4096 (k_lu_factor_big):
// what bounds it is the LDS that holds L21^T and U12 of a panel, 2 * PB * N elements
inline int max_rows(int dtype) { return dtype == LQP_F32 ? 4096 : 2048; }
constexpr size_t kAlign = 256;
constexpr int kSplitMaxB = 8192;     // two-workgroup loop: exchange granules (32 KB per problem) are carved for batches up to this

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename U> U* take(size_t count) {
        off = (off + kAlign - 1) / kAlign * kAlign;
        U* r = base ? (U*)(base + off) : nullptr;
        off += count * sizeof(U);
        return r;
    }
};

#define HIP_OK(call)                                                      \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) {                                           \
            if (getenv("LQP_DEBUG")) fprintf(stderr, "[lqp] %s failed: %s\n", #call, hipGetErrorString(e_)); \
            return LQP_ERR_HIP;                                           \
        }                                                                 \
    } while (0)

// dynamic LDS above 64 KB needs an opt-in per kernel
std::mutex g_attr_mutex;
std::unordered_map<const void*, int> g_attr_set;
int ensure_lds(const void* fn, int bytes) {
    if (bytes > 160 * 1024) return LQP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    auto it = g_attr_set.find(fn);
    if (it != g_attr_set.end() && it->second >= bytes) return LQP_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return LQP_ERR_HIP;
    g_attr_set[fn] = bytes;
    return LQP_OK;
}

// Occupancy answers and CU counts do not change for a (kernel, block size, LDS) on a device: the runtime calls cost
// microseconds each and a forward used to make nine of them.
struct OccKey { const void* fn; int nt, lds, dev; bool operator==(const OccKey& o) const { return fn == o.fn && nt == o.nt && lds == o.lds && dev == o.dev; } };
struct OccHash { size_t operator()(const OccKey& k) const { return std::hash<const void*>()(k.fn) ^ ((size_t)k.nt * 1315423911u) ^ ((size_t)k.lds << 7) ^ (size_t)k.dev; } };
std::unordered_map<OccKey, int, OccHash> g_occ;
int g_cus[64];
bool current_device_cus(int* dev, int* cus) {
    if (hipGetDevice(dev) != hipSuccess || *dev < 0 || *dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    if (g_cus[*dev] == 0 && hipDeviceGetAttribute(&g_cus[*dev], hipDeviceAttributeMultiprocessorCount, *dev) != hipSuccess) return false;
    *cus = g_cus[*dev];
    return true;
}
template <typename F> bool blocks_per_cu(int* per_cu, F fn, int nt, int lds, int dev) {
    const OccKey key{(const void*)fn, nt, lds, dev};
    {
        std::lock_guard<std::mutex> lock(g_attr_mutex);
        auto it = g_occ.find(key);
        if (it != g_occ.end()) { *per_cu = it->second; return true; }
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, fn, nt, lds) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    g_occ[key] = *per_cu;
    return true;
}

// ---- optional per-kernel-class timing with HIP events on the launch stream ----
enum { PC_SETUP = 0, PC_LU, PC_PACK, PC_LOOP, PC_RHO, PC_EPILOGUE, PC_BWD_BUILD, PC_SOLVE, PC_BWD_EPILOGUE,
       PC_MISC, PC_LOOP_TAIL, PC_SPD_INV, PC_EQ_CORR, PC_BWD_CHOL, PC_UNROLL, PC_UNROLL_SCALE, PC_COUNT };
struct ProfRec { int cls; hipEvent_t a, b; };
std::mutex g_prof_mutex;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
double g_prof_ms[PC_COUNT];
long long g_prof_n[PC_COUNT];

struct ProfScope {
    hipStream_t st; int cls; bool on; hipEvent_t a, b;
    ProfScope(hipStream_t s, int c) : st(s), cls(c), on(g_prof_on) {
        if (on) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(b, st);
            std::lock_guard<std::mutex> lock(g_prof_mutex);
            g_prof_recs.push_back({cls, a, b});
        }
    }
};

// (environment knobs are read once: a forward consults two dozen of them; tests that flip one do so in a fresh process
//  or through LQP_ENV_NOCACHE=1)
int env_int(const char* name, int dflt) {
    static const bool nocache = getenv("LQP_ENV_NOCACHE") != nullptr;
    if (nocache) { const char* s = getenv(name); return s ? atoi(s) : dflt; }
    static std::mutex m;
    static std::unordered_map<std::string, std::pair<bool, int>> cache;
    std::lock_guard<std::mutex> lock(m);
    auto it = cache.find(name);
    if (it == cache.end()) {
        const char* s = getenv(name);
        it = cache.emplace(name, std::make_pair(s != nullptr, s ? atoi(s) : 0)).first;
    }
    return it->second.first ? it->second.second : dflt;
}

// Every LQP_* environment switch that steers a schedule, read ONCE per process (tests that flip one run in a fresh process
// or set LQP_ENV_NOCACHE=1, which re-reads them at every use).  -1: not set, the library decides.
struct Knobs {
    int bwd_chol;              // LQP_BWD_CHOL
    int bwd_early;             // LQP_BWD_EARLY
    int bwd_full;              // LQP_BWD_FULL
    int bwd_lookahead;         // LQP_BWD_LOOKAHEAD
    int inv_xcd;               // LQP_INV_XCD: the inverse's column tiles of one problem on one XCD (0: grid order)
    int dbg_lu2_absent;        // LQP_DBG_LU2_ABSENT: tests only -- the partner workgroups of the two-workgroup LU are not launched
    int dbg_loop_absent;       // LQP_DBG_LOOP_ABSENT: tests only -- bit 0 ... of the two-workgroup loop, bit 1 ... of the resident sweep, bit 2 ... of the unroll sweep, bit 3 ... of the two-workgroup streaming loop
    int bwd_refine;            // LQP_BWD_REFINE
    int dbg_qpass;             // LQP_DBG_QPASS
    int dbg_setup;             // LQP_DBG_SETUP
    int epi_slabs;             // LQP_EPI_SLABS
    int eq_in_loop;            // LQP_EQ_IN_LOOP
    int launch_mode;           // LQP_LAUNCH_MODE
    int linsolve;              // LQP_LINSOLVE
    int loop512;               // LQP_LOOP512
    int loop_dense;            // LQP_LOOP_DENSE
    int loop_dense_w;          // LQP_LOOP_DENSE_W
    int loop_small;            // LQP_LOOP_SMALL
    int loop_split;            // LQP_LOOP_SPLIT
    int loop_split4;           // LQP_LOOP_SPLIT4
    int loop_np2;              // LQP_LOOP_NP2
    int spd_big_fuse;          // LQP_SPD_BIG_FUSE
    int spd_big_f16;           // LQP_SPD_BIG_F16
    int loop_split_seg;        // LQP_LOOP_SPLIT_SEG
    int lu2;                   // LQP_LU2
    int lu_wide;               // LQP_LU_WIDE
    int unroll_split;          // LQP_UNROLL_SPLIT
    int lu_wide_min;           // LQP_LU_WIDE_MIN: the wide LU above this many rows (experiments; default 1024)
    int lu_mfma;               // LQP_LU_MFMA
    int lu_nt;                 // LQP_LU_NT
    int lu_pb;                 // LQP_LU_PB
    int nosync_max_events;     // LQP_NOSYNC_MAX_EVENTS
    int prep_fused;            // LQP_PREP_FUSED
    int prep_one;              // LQP_PREP_ONE
    int qpass;                 // LQP_QPASS
    int qs_lazy;               // LQP_QS_LAZY
    int resident;              // LQP_RESIDENT
    int rho_late;              // LQP_RHO_LATE
    int spd_big;               // LQP_SPD_BIG
    int spd_ptasks;            // LQP_SPD_PTASKS
    int spd_resident;          // LQP_SPD_RESIDENT
    int spd_resident4;         // LQP_SPD_RESIDENT4
    int hot_past;              // LQP_HOT_PAST: the persistent two-workgroup loop runs on past rho events that change nothing
    int hot_rounds;            // LQP_HOT_ROUNDS: with a GIVEN rho, rounds of {rho update, gated refactorisation, hot loop again} enqueued behind the first hot launch
    int bwd_equil;             // LQP_BWD_EQUIL: power-of-two symmetric equilibration of the Cholesky backward's free-set block
    int bwd_f16;               // LQP_BWD_F16: the backward's look-ahead Cholesky with its tile products on the float16 pipe (with LQP_SPD_F16)
    int spd_turns;             // LQP_SPD_TURNS: more matrices than half the CUs -> the resident sweep anyway, its pairs taking turns on the chip
    int spd_f16;               // LQP_SPD_F16: the resident sweep's panel products on the float16 matrix pipe (two-half operands); 0: float32 matrix instructions
    int spd_split;             // LQP_SPD_SPLIT
    int spec_launches;         // LQP_SPEC_LAUNCHES
    int split2;                // LQP_SPLIT2
    int sym512;                // LQP_SYM512
    int sync_plan;             // LQP_SYNC_PLAN
    int tail_epilogue;         // LQP_TAIL_EPILOGUE
    int xcd_local;             // LQP_XCD_LOCAL
};
Knobs read_knobs() {
    Knobs k;
    k.bwd_chol = env_int("LQP_BWD_CHOL", 1);
    k.bwd_early = env_int("LQP_BWD_EARLY", 1);
    k.bwd_full = env_int("LQP_BWD_FULL", 0);
    k.bwd_lookahead = env_int("LQP_BWD_LOOKAHEAD", 1);
    k.inv_xcd = env_int("LQP_INV_XCD", 1);
    k.dbg_lu2_absent = env_int("LQP_DBG_LU2_ABSENT", 0);
    k.dbg_loop_absent = env_int("LQP_DBG_LOOP_ABSENT", 0);
    k.bwd_refine = env_int("LQP_BWD_REFINE", 1);
    k.dbg_qpass = env_int("LQP_DBG_QPASS", 0);
    k.dbg_setup = env_int("LQP_DBG_SETUP", 0);
    k.epi_slabs = env_int("LQP_EPI_SLABS", -1);
    k.eq_in_loop = env_int("LQP_EQ_IN_LOOP", 1);
    k.launch_mode = env_int("LQP_LAUNCH_MODE", 2);
    k.linsolve = env_int("LQP_LINSOLVE", 0);
    k.loop512 = env_int("LQP_LOOP512", 0);
    k.loop_dense = env_int("LQP_LOOP_DENSE", 1);
    k.loop_dense_w = env_int("LQP_LOOP_DENSE_W", 1);
    k.loop_small = env_int("LQP_LOOP_SMALL", 1);
    k.loop_split = env_int("LQP_LOOP_SPLIT", 1);
    k.loop_split4 = env_int("LQP_LOOP_SPLIT4", 1);
    k.loop_np2 = env_int("LQP_LOOP_NP2", 1);
    k.spd_big_fuse = env_int("LQP_SPD_BIG_FUSE", 1);
    k.spd_big_f16 = env_int("LQP_SPD_BIG_F16", 1);
    k.loop_split_seg = env_int("LQP_LOOP_SPLIT_SEG", 1);
    k.lu2 = env_int("LQP_LU2", 1);
    k.lu_wide = env_int("LQP_LU_WIDE", 1);
    k.unroll_split = env_int("LQP_UNROLL_SPLIT", 1);
    k.lu_wide_min = env_int("LQP_LU_WIDE_MIN", 1024);
    k.lu_mfma = env_int("LQP_LU_MFMA", 1);
    k.lu_nt = env_int("LQP_LU_NT", 0);
    k.lu_pb = env_int("LQP_LU_PB", 0);
    k.nosync_max_events = env_int("LQP_NOSYNC_MAX_EVENTS", 12);
    k.prep_fused = env_int("LQP_PREP_FUSED", 1);
    k.prep_one = env_int("LQP_PREP_ONE", 1);
    k.qpass = env_int("LQP_QPASS", 1);
    k.qs_lazy = env_int("LQP_QS_LAZY", 1);
    k.resident = env_int("LQP_RESIDENT", 1);
    k.rho_late = env_int("LQP_RHO_LATE", 1);
    k.spd_big = env_int("LQP_SPD_BIG", 1);
    k.spd_ptasks = env_int("LQP_SPD_PTASKS", 48);
    k.spd_resident = env_int("LQP_SPD_RESIDENT", 1);
    k.spd_resident4 = env_int("LQP_SPD_RESIDENT4", 1);
    k.spd_f16 = env_int("LQP_SPD_F16", 1);
    k.spd_turns = env_int("LQP_SPD_TURNS", 1);
    k.bwd_f16 = env_int("LQP_BWD_F16", 1);
    k.bwd_equil = env_int("LQP_BWD_EQUIL", 1);
    k.hot_past = env_int("LQP_HOT_PAST", 1);
    k.hot_rounds = env_int("LQP_HOT_ROUNDS", 2);
    k.spd_split = env_int("LQP_SPD_SPLIT", -1);
    k.spec_launches = env_int("LQP_SPEC_LAUNCHES", 6);
    k.split2 = env_int("LQP_SPLIT2", 1);
    k.sym512 = env_int("LQP_SYM512", 0);
    k.sync_plan = env_int("LQP_SYNC_PLAN", 1);
    k.tail_epilogue = env_int("LQP_TAIL_EPILOGUE", 1);
    k.xcd_local = env_int("LQP_XCD_LOCAL", 1);
    return k;
}
const Knobs& knobs() {
    static const bool nocache = getenv("LQP_ENV_NOCACHE") != nullptr;
    static const Knobs cached = read_knobs();
    if (!nocache) return cached;
    static thread_local Knobs fresh;
    fresh = read_knobs();
    return fresh;
}

// The LU kernels that share a matrix between workgroups (lqp_lu2.hpp, lqp_lu_wide.hpp) need all of them resident at once; the
// launch checks that they fit, not that the chip is otherwise idle (another stream's kernels, RCCL).  When a hand-off times out
// (info word -7: the waits are bounded, the kernels drain) the call is repeated ONCE with one workgroup per matrix.
thread_local bool t_single_wg_lu = false;
struct SingleWgLu {
    SingleWgLu() { t_single_wg_lu = true; }
    ~SingleWgLu() { t_single_wg_lu = false; }
};
unsigned long long* g_lu_dbg = nullptr;     // optional device buffer (4 counters per problem), debug only

// ---- LU launch: pick panel width / trailing-update flavour --------------------
template <typename T, int PB, bool MFMA, int NT>
int launch_lu_impl(hipStream_t st, int B, T* M, int N, int ld, size_t mstride, int* piv, int pstride, int* info,
                   const int* gate, const int* nvec) {
    const int lds = LuLds<T, PB>(round_up(N, 64)).total;
    auto fn = k_lu_factor<T, PB, MFMA, NT>;
    int rc = ensure_lds((const void*)fn, lds);
    if (rc) return rc;
    { ProfScope ps(st, PC_LU);
      hipLaunchKernelGGL(fn, dim3(B), dim3(NT), lds, st, M, N, ld, mstride, piv, pstride, info, gate, knobs().dbg_setup ? nullptr : g_lu_dbg, nvec); }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

// N <= 512: 512 threads (one row per thread, 256-VGPR budget) -> 32-column panel for f32, 16 for f64;
// larger N: 1024 threads, 16 / 8 columns.  LQP_LU_PB / LQP_LU_MFMA / LQP_LU_NT override for experiments.
// 1024 < N <= 2048: two panel rows per thread (k_lu_factor_big)
template <typename T>
int launch_lu_big(hipStream_t st, T* M, int B, int N, int ld, size_t mstride, int* piv, int pstride, int* info,
                  const int* gate, const int* nvec) {
    if (N > (sizeof(T) == 4 ? 4096 : 2048)) return LQP_ERR_UNSUPPORTED;
    const int lds = N > 2048 ? LuLds<T, 4>(round_up(N, 64)).total : LuLds<T, lu_big_panel<T>()>(round_up(N, 64)).total;
    auto fn = k_lu_factor_big<T>;
    const int rc = ensure_lds((const void*)fn, lds);
    if (rc) return rc;
    ProfScope ps(st, PC_LU);
    hipLaunchKernelGGL(fn, dim3(B), dim3(LQP_NT), lds, st, M, N, ld, mstride, piv, pstride, info, gate, nvec);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

// Two workgroups per matrix (lqp_lu2.hpp) when the batch leaves half the chip idle: N <= 512, at least three column blocks, 2 B
// workgroups resident (they wait for each other inside the launch), and the caller has 8 * LU2_SCR_WORDS bytes of scratch per
// problem (`scr`, problem b at scr + b * scr_stride words).  Returns -1: not applicable, take the one-workgroup kernel.
std::atomic<unsigned int> g_lu2_epoch{(unsigned int)std::chrono::steady_clock::now().time_since_epoch().count() | 1u};
template <typename T>
int launch_lu2(hipStream_t st, T* M, int B, int N, int ld, size_t mstride, int* piv, int pstride, int* info,
               const int* gate, const int* nvec, unsigned long long* scr, size_t scr_stride) {
    constexpr int PB = lu2_panel_width<T>();
    if (!scr || scr_stride < (size_t)LU2_SCR_WORDS || N > 512 || N < 3 * PB || knobs().lu2 == 0 || t_single_wg_lu) return -1;
    int dev = 0, cus = 0, per_cu = 0;
    auto fn = k_lu_factor2<T, PB>;
    const int lds = Lu2Lds<T, PB>(round_up(N, 64)).total;
    if (!current_device_cus(&dev, &cus) || ensure_lds((const void*)fn, lds) != LQP_OK ||
        !blocks_per_cu(&per_cu, fn, LU2_NT, lds, dev) || per_cu < 1 || shared_grid(B, 2) > cus * per_cu)
        return -1;
    const unsigned int epoch = g_lu2_epoch.fetch_add(1u) + 1u;
    { ProfScope ps(st, PC_LU);
      // (LQP_DBG_LU2_ABSENT: the hand-offs of workgroups b time out, info = -7, the caller repeats on one workgroup per matrix)
      hipLaunchKernelGGL(fn, dim3(knobs().dbg_lu2_absent ? B : shared_grid(B, 2)), dim3(LU2_NT), lds, st, M, N, ld, mstride, piv, pstride, info, gate, nvec, scr,
                         scr_stride, epoch, knobs().dbg_setup ? nullptr : g_lu_dbg, B, knobs().xcd_local != 0 ? 1 : 0); }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

// Several workgroups per matrix above 1024 rows (lqp_lu_wide.hpp) when the batch leaves most of the chip idle: W = what fits
// resident / B workgroups (>= 2) own the 128-byte column tiles cyclically.  `scr`: 4 * luw_scratch_words<T>(N) bytes per problem.
// Returns -1: not applicable, take the one-workgroup kernel.
template <typename T>
int launch_lu_wide(hipStream_t st, T* M, int B, int N, int ld, size_t mstride, int* piv, int pstride, int* info,
                   const int* gate, const int* nvec, unsigned long long* scr, size_t scr_stride) {
    const bool tall = N > 2048;          // float32 only: sixteen panel rows per thread, panels of 4 columns
    if (!scr || t_single_wg_lu || 2 * scr_stride < luw_scratch_words<T>(N) || N <= knobs().lu_wide_min || N > (sizeof(T) == 4 ? 4096 : 2048) ||
        knobs().lu_wide == 0 || (ld % 32) != 0 || (mstride % 32) != 0 || (((uintptr_t)M) % 128) != 0)
        return -1;
    int dev = 0, cus = 0, per_cu = 0;
    auto fn = k_lu_factor_wide<T>;
    if constexpr (sizeof(T) == 4) { if (tall) fn = k_lu_factor_wide_tall<T>; }
    const int lds = tall ? LuLds<T, 4>(round_up(N, 64)).total : LuLds<T, luw_pb<T>()>(round_up(N, 64)).total;
    if (!current_device_cus(&dev, &cus) || ensure_lds((const void*)fn, lds) != LQP_OK ||
        !blocks_per_cu(&per_cu, fn, LQP_NT, lds, dev) || per_cu < 1)
        return -1;
    int W = (cus * per_cu) / B;
    const int ntile = (N + luw_tw<T>() - 1) / luw_tw<T>();
    if (W > ntile) W = ntile;
    if (W > LUW_HDR - 2) W = LUW_HDR - 2;
    if (W < 2) return -1;
    const unsigned int epoch = g_lu2_epoch.fetch_add(1u) + 1u;
    { ProfScope ps(st, PC_LU);
      hipLaunchKernelGGL(fn, dim3(W * B), dim3(LQP_NT), lds, st, M, N, ld, mstride, piv, pstride, info, gate, nvec, (int*)scr,
                         2 * scr_stride, epoch, B, knobs().dbg_setup ? nullptr : g_lu_dbg); }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

int launch_lu(hipStream_t st, float* M, int B, int N, int ld, size_t mstride, int* piv, int pstride, int* info,
              const int* gate, const int* nvec = nullptr, unsigned long long* scr = nullptr, size_t scr_stride = 0) {
    if (N > 512) {
        const int rw = launch_lu_wide<float>(st, M, B, N, ld, mstride, piv, pstride, info, gate, nvec, scr, scr_stride);
        if (rw >= 0) return rw;
    }
    if (N > 1024) return launch_lu_big<float>(st, M, B, N, ld, mstride, piv, pstride, info, gate, nvec);
    { const int r2 = launch_lu2<float>(st, M, B, N, ld, mstride, piv, pstride, info, gate, nvec, scr, scr_stride);
      if (r2 >= 0) return r2; }
    int nt = lu_threads<float>(N);
    int pb = lu_panel_width<float>(N);
    const int want_nt = knobs().lu_nt;
    if (want_nt == 1024) { nt = 1024; pb = std::min(pb, 16); }
    const int want = knobs().lu_pb;
    if ((want == 8 || want == 16 || (want == 32 && nt == 512)) && 2 * want * round_up(N, 64) * 4 <= 128 * 1024) pb = want;
    const bool mfma = knobs().lu_mfma != 0;
#define LQP_LU_CASE(PBV, MF, NTV) return launch_lu_impl<float, PBV, MF, NTV>(st, B, M, N, ld, mstride, piv, pstride, info, gate, nvec)
    if (nt == 512) {
        if (pb == 32) { if (mfma) LQP_LU_CASE(32, true, 512); LQP_LU_CASE(32, false, 512); }
        if (pb == 16) { if (mfma) LQP_LU_CASE(16, true, 512); LQP_LU_CASE(16, false, 512); }
        LQP_LU_CASE(8, false, 512);
    }
    if (pb == 16) { if (mfma) LQP_LU_CASE(16, true, 1024); LQP_LU_CASE(16, false, 1024); }
    LQP_LU_CASE(8, false, 1024);
#undef LQP_LU_CASE
}
int launch_lu(hipStream_t st, double* M, int B, int N, int ld, size_t mstride, int* piv, int pstride, int* info,
              const int* gate, const int* nvec = nullptr, unsigned long long* scr = nullptr, size_t scr_stride = 0) {
    if (N > 512) {
        const int rw = launch_lu_wide<double>(st, M, B, N, ld, mstride, piv, pstride, info, gate, nvec, scr, scr_stride);
        if (rw >= 0) return rw;
    }
    if (N > 1024) return launch_lu_big<double>(st, M, B, N, ld, mstride, piv, pstride, info, gate, nvec);
    { const int r2 = launch_lu2<double>(st, M, B, N, ld, mstride, piv, pstride, info, gate, nvec, scr, scr_stride);
      if (r2 >= 0) return r2; }
    const int nt = lu_threads<double>(N);
    const int pb = lu_panel_width<double>(N);
#define LQP_LU_CASE(PBV, NTV) return launch_lu_impl<double, PBV, true, NTV>(st, B, M, N, ld, mstride, piv, pstride, info, gate, nvec)
    if (nt == 512) {
        if (pb == 16) LQP_LU_CASE(16, 512);
        LQP_LU_CASE(8, 512);
    }
    if (pb == 16) LQP_LU_CASE(16, 1024);
    LQP_LU_CASE(8, 1024);
#undef LQP_LU_CASE
}

template <typename T>
int launch_pack(hipStream_t st, int B, const T* LU, int N, int ld, size_t mstride, const int* piv, int pstride,
                T* packed, int* dest, const int* gate, const int* nvec = nullptr) {
    const int K = round_up(N, LQP_NB) / LQP_NB;
    const int lds = pack_lds_bytes<T>();
    auto fn = k_pack<T>;
    int rc = ensure_lds((const void*)fn, lds);
    if (rc) return rc;
    const bool vec_ok = (ld % 4 == 0) && (mstride % 4 == 0) && (((uintptr_t)LU) % (4 * sizeof(T)) == 0);
    ProfScope ps(st, PC_PACK);
    const int split = (B <= 128 && knobs().split2) ? 2 : 1;      // use the idle half of the chip
    hipLaunchKernelGGL(fn, dim3(B, split), dim3(LQP_NT), lds, st, LU, N, ld, mstride, piv, pstride, packed,
                       packed_blocks(K) * LQP_BLK, dest, K * LQP_NB, vec_ok ? 1 : 0, gate, nvec);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

template <typename T>
int launch_solve(hipStream_t st, int B, const T* packed, int N, const int* dest, T* rhs, int nrhs, size_t bstride,
                 int rstride, int cstride, const int* nvec = nullptr) {
    const int K = round_up(N, LQP_NB) / LQP_NB, Np = K * LQP_NB;
    const int lds = solve_lds_bytes<T>(Np);
    auto fn = k_packed_solve<T>;
    int rc = ensure_lds((const void*)fn, lds);
    if (rc) return rc;
    ProfScope ps(st, PC_SOLVE);
    hipLaunchKernelGGL(fn, dim3(B), dim3(LQP_NT), lds, st, packed, N, Np, K, dest, rhs, nrhs, bstride, rstride, cstride, nvec);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

// X = M^-1 from the packed factor (lqp_dense.hpp): G workgroups per matrix, each takes column tiles g, g + G, ...
template <typename T, bool SMALL>
int launch_lu_inverse_cfg(hipStream_t st, int B, int N, const T* packed, size_t pkstride, const int* dest, int dstride, T* X,
                          size_t xstride, int ldx, const int* gate) {
    typedef InvCfg<T, SMALL> C;
    const int Np = round_up(N, LQP_NB);
    const int lds = lu_inverse_lds_bytes<T, SMALL>(Np);
    if (lds > 160 * 1024) return LQP_ERR_UNSUPPORTED;
    auto fn = k_lu_inverse<T, SMALL>;
    const int rc = ensure_lds((const void*)fn, lds);
    if (rc) return rc;
    int dev = 0, cus = 256;
    (void)current_device_cus(&dev, &cus);
    const int ntiles = (N + C::TWG - 1) / C::TWG;
    // (column tiles are independent: every tile its own workgroup while the grid stays within a few waves of workgroups per CU --
    //  several 256-thread workgroups per CU hide each other's barriers and operand loads)
    int G = std::max(1, std::min(ntiles, (12 * cus) / std::max(B, 1)));
    ProfScope ps(st, PC_PACK);
    // (the tiles of a problem side by side on one XCD -- they all read the same factor --, see k_lu_inverse)
    if (G > 1 && G < 65536 && B < 32768 && knobs().inv_xcd != 0)
        hipLaunchKernelGGL(fn, dim3(8 * ((B + 7) / 8) * G), dim3(256), lds, st, packed, pkstride, N, G | (B << 16), dest, dstride, X, xstride,
                           ldx, gate);
    else
        hipLaunchKernelGGL(fn, dim3(B, G), dim3(256), lds, st, packed, pkstride, N, 0, dest, dstride, X, xstride, ldx, gate);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}
// float32: 64-column tiles on v_mfma_f32_32x32x2 while Y fits the LDS (N <= 576), 16-column tiles on v_mfma_f32_16x16x4 above
template <typename T>
int launch_lu_inverse(hipStream_t st, int B, int N, const T* packed, size_t pkstride, const int* dest, int dstride, T* X,
                      size_t xstride, int ldx, const int* gate) {
    if constexpr (sizeof(T) == 4) {
        if (lu_inverse_lds_bytes<T, false>(round_up(N, LQP_NB)) > 160 * 1024)
            return launch_lu_inverse_cfg<T, true>(st, B, N, packed, pkstride, dest, dstride, X, xstride, ldx, gate);
    }
    return launch_lu_inverse_cfg<T, false>(st, B, N, packed, pkstride, dest, dstride, X, xstride, ldx, gate);
}
template <typename T> inline bool lu_inverse_fits(int Np) {
    return lu_inverse_lds_bytes<T, false>(Np) <= 160 * 1024 || (sizeof(T) == 4 && lu_inverse_lds_bytes<T, true>(Np) <= 160 * 1024);
}

// first failing batch index from the per-problem info array (host side, after a sync)
// host_info: the same words in pinned host memory, stored there by the last kernel the stream has run (host_report): the
// stream is waited for, nothing is copied
int first_failure(hipStream_t st, const int* info_dev, int B, int* fail_index, const int* host_info = nullptr) {
    std::vector<int> h;
    const int* words = host_info;
    if (host_info) {
        HIP_OK(hipStreamSynchronize(st));
    } else {
        h.resize(B);
        HIP_OK(hipMemcpyAsync(h.data(), info_dev, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        words = h.data();
    }
    *fail_index = -1;
    for (int i = 0; i < B; ++i)
        if (((const volatile int*)words)[i] == -7) { *fail_index = i; return LQP_ERR_TIMEOUT; }      // a multi-workgroup LU never met its partner
    for (int i = 0; i < B; ++i)
        if (((const volatile int*)words)[i] != 0) { *fail_index = i; return LQP_ERR_SINGULAR; }
    return LQP_OK;
}


// A synchronous call waits for its report, not for its stream: the words of a host report (pinned host memory, all set to
// -1 by the call before its first launch; no word the kernels store is -1) are polled until none is missing.  A
// hipStreamSynchronize of a ~0.5 ms schedule parks the thread after 100 us of spinning and pays the interrupt + wake-up
// on top of the completion signal's trip; the polled word is seen ~a microsecond after the store.  The device results
// themselves are stream-ordered like those of any torch operator -- the host only ever reads the report.  The stream is
// queried every ~50 us of waiting: a drained (or failed) stream whose report is still incomplete is an error, not a hang.
void report_reset(int* host_report, int words) {
    for (int i = 0; i < words; ++i) ((volatile int*)host_report)[i] = -1;
}
int wait_report(hipStream_t st, const int* host_report, int words) {
    const volatile int* r = (const volatile int*)host_report;
    int first_missing = 0;
    auto t_query = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        while (first_missing < words && r[first_missing] != -1) ++first_missing;      // (-7, a hand-off that timed out, HAS arrived)
        if (first_missing >= words) { std::atomic_thread_fence(std::memory_order_acquire); return LQP_OK; }
        __builtin_ia32_pause();
        if ((spins & 255u) == 255u) {
            const auto now = std::chrono::steady_clock::now();
            if (now - t_query > std::chrono::microseconds(50)) {
                t_query = now;
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {              // everything enqueued has run: the report must be complete now
                    for (int i = first_missing; i < words; ++i) if (r[i] == -1) return LQP_ERR_HIP;
                    return LQP_OK;
                }
                if (q != hipErrorNotReady) return LQP_ERR_HIP;
            }
        }
    }
}

// What a synchronous forward does once its whole schedule is enqueued: wait for the report (wait_report, when it sits in
// the caller's pinned memory), look at the info / flag words of every problem, fill the statistics.  `stats` comes in
// with n_launch / linsolve_used / factor_launches / loop_workgroups of the enqueue and keeps them.
// LQP_RETRY_LU (internal): the matrix left the symmetric x-update -- repeat the solve with linsolve = 1.
constexpr int LQP_RETRY_LU = 100;
inline bool flags_timeout_loop(const int* rep) { return ((const volatile int*)rep)[ST_TIMEOUT] != 0; }
int collect_report(hipStream_t st, const int* rep, const bool polled, const int B, const int max_iters, const int check,
                   lqp_boxqp_stats* stats) {
    if (polled) {
        const int rc = wait_report(st, rep, ST_WORDS + 2 * B);
        if (rc) return rc;
    }
    const volatile int* rv = (const volatile int*)rep;
    int fail_index = -1, flags_or = 0;
    bool lu_timeout = false;
    for (int i = 0; i < B; ++i) {
        if (fail_index < 0 && rv[ST_WORDS + i] != 0) fail_index = i;
        lu_timeout = lu_timeout || rv[ST_WORDS + i] == -7;          // (a multi-workgroup LU never met its partner)
        flags_or |= rv[ST_WORDS + B + i];
    }
    if (lu_timeout && stats->linsolve_used != 2) return LQP_ERR_TIMEOUT;
    if (fail_index >= 0 && stats->linsolve_used == 2) return LQP_RETRY_LU;
    if (fail_index >= 0) {
        memset(stats, 0, sizeof(*stats));
        stats->fail_index = fail_index;
        return LQP_ERR_SINGULAR;
    }
    if (rv[ST_TIMEOUT] || (flags_or & RP_TIMEOUT)) return LQP_ERR_TIMEOUT;
    const int final_iter = rv[ST_DONE] ? rv[ST_FINAL_ITER] : max_iters - 1;
    stats->iters = final_iter;
    stats->n_factor = 1 + rv[ST_NFACTOR];
    stats->n_solve = final_iter + 1;
    stats->n_check = final_iter / check + 1;
    stats->rho_updated = rv[ST_RHO_UPDATED];
    stats->fail_index = -1;
    stats->mode_used = 2;
    stats->any_lb = rv[ST_ANY_LB]; stats->any_ub = rv[ST_ANY_UB];
    return LQP_OK;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T> struct FwdLayout {
    FwdParams<T> P;
    size_t bytes;
    unsigned int* vtrace_area;      // [kRing][2]: the check trace (FwdParams::vtrace points here when ctrl.reserved2 bit 2 asks for it)
};

template <typename T>
FwdLayout<T> carve_forward(void* ws, int B, int n, int m) {
    FwdLayout<T> L;
    memset(&L.P, 0, sizeof(L.P));
    FwdParams<T>& P = L.P;
    P.B = B; P.n = n; P.m = m; P.N = n + m;
    P.Np = round_up(P.N, LQP_NB); P.K = P.Np / LQP_NB; P.ldq = round_up(n, 4);
    P.Ks = round_up(n, LQP_NB) / LQP_NB;
    P.vstride = vec_stride(n, m);
    Carver c(ws);
    P.status = c.take<int>(ST_WORDS);
    P.counters = c.take<unsigned int>((size_t)kRing * CT_WORDS);
    P.info = c.take<int>(B);            // (status | counters | info: one contiguous region for the deferred error fetch)
    P.bflags = c.take<int>(B);
    L.vtrace_area = c.take<unsigned int>((size_t)kRing * 2);      // (the check trace of verbose=True; in use: FwdParams::vtrace)
    P.scal = c.take<T>((size_t)B * SC_WORDS);
    P.vecs = c.take<T>((size_t)B * P.vstride);
    P.piv = c.take<int>((size_t)B * P.Np);
    P.dest = c.take<int>((size_t)B * P.Np);
    P.Qs = c.take<T>((size_t)B * n * P.ldq);
    P.M = c.take<T>((size_t)B * P.Np * P.Np);
    P.packed = c.take<T>((size_t)B * packed_blocks(P.K) * LQP_BLK);
    // granules of the two-workgroup loop (only ever used when 2 B workgroups fit the chip)
    // (only where a two-workgroup kernel can run at all: 32 KB per problem -- a batch of 8192 small problems used to carry 256 MB of it)
    // (above 512 rows: the streaming loop on two workgroups per problem, k_admm_loop_np2 -- batches that can leave half the chip idle)
    P.xchg = (sizeof(T) == 4 && P.Ks >= SPLIT_MINK && ((B <= kSplitMaxB && P.Ks <= SPD_MAXK) || (B <= 256 && P.Ks <= SPD_BIGK)))
                 ? c.take<unsigned long long>((size_t)B * (XCHG_WORDS + XCHG_TAIL)) : nullptr;
    // granules of the dense LU-tier loop (lqp_dense.hpp): batches that can leave half a chip idle, n <= 256
    // granules of the dense LU-tier loops: two workgroups per problem to n = 256; W workgroups per problem for small batches at any
    // n the inverse kernel takes (two parities x n elements x one or two words)
    P.dnx_words = (B <= 256 && n <= DENSE_NMAX && P.N <= 1024) ? DNX_WORDS
                : (B <= 64 && P.N <= 2048) ? (int)std::max((size_t)DNX_WORDS, densew_xchg_words<T>(n)) : 0;
    P.dnx = P.dnx_words ? c.take<unsigned long long>((size_t)B * P.dnx_words) : nullptr;
    L.bytes = c.off + kAlign;
    return L;
}

template <typename T>
int forward_impl(hipStream_t st, int B, int n, int m, const void* Q, const void* p, const void* A, const void* b,
                 const void* lb, const void* ub, const lqp_boxqp_ctrl* ctl, const void* rho_in, void* x, void* z,
                 void* u, void* lams, void* nus, void* rho_out, lqp_boxqp_stats* stats, void* ws, size_t ws_bytes,
                 const int retry = 0) {
    // retry: bit 0 -- the symmetric x-update gave the solve up (not symmetric / not positive definite in f32): pivoted LU;
    //        bit 1 -- a pair of the turn-taking two-workgroup loop (split_seg) waited for its partner in vain: the one-workgroup loop
    //        bit 2 -- a kernel that shares its problems between workgroups gave up waiting (bounded spins: something else holds the
    //                 CUs -- another stream, another process, RCCL): nothing shared this time -- one workgroup per matrix in every
    //                 factorisation, host-launched check segments instead of the persistent loop with its grid barrier
    const bool force_lu = (retry & 1) != 0;
    const bool solo = (retry & 4) != 0;
    struct SoloScope { bool on; explicit SoloScope(bool o) : on(o && !t_single_wg_lu) { if (on) t_single_wg_lu = true; }
                       ~SoloScope() { if (on) t_single_wg_lu = false; } } solo_scope(solo);
    FwdLayout<T> L = carve_forward<T>(ws, B, n, m);
    if (ws_bytes < L.bytes) return LQP_ERR_WORKSPACE;
    FwdParams<T>& P = L.P;
    P.Q = (const T*)Q; P.p = (const T*)p; P.A = (const T*)A; P.b = (const T*)b;
    P.lb = (const T*)lb; P.ub = (const T*)ub; P.rho_in = (const T*)rho_in; P.beta_in = (const T*)ctl->beta_in;
    P.x = (T*)x; P.z = (T*)z; P.u = (T*)u; P.lams = (T*)lams; P.nus = (T*)nus; P.rho_out = (T*)rho_out;
    P.scale = ctl->scale; P.bound_flags_in = (const int*)ctl->bound_flags_in;
    if (ctl->reserved2 & 4) {           // verbose: the largest primal / dual error of the batch at every check (lqp_boxqp_check_trace)
        P.vtrace = L.vtrace_area;
        HIP_OK(hipMemsetAsync(P.vtrace, 0, sizeof(unsigned int) * 2 * kRing, st));
    }
    P.host_report = (int*)ctl->host_report;
    if (P.host_report) report_reset(P.host_report, ST_WORDS + 2 * B);      // (before the first launch: see wait_report)
    P.xcd_local = knobs().xcd_local != 0 ? 1 : 0;
    P.dbg_qpass = knobs().dbg_qpass;      // (bit 0: the second half of the debug buffer; bits 8..: the wave whose stamps the resident sweep records)
    P.zero_words = (int)(((char*)(P.counters + (size_t)kRing * CT_WORDS) - (char*)P.status) / sizeof(int));
    P.dbg = g_lu_dbg;
    P.dbg_setup = nullptr;
    if (g_lu_dbg && knobs().dbg_setup) { P.dbg_setup = g_lu_dbg; P.dbg = nullptr; }
    P.rho_mode = ctl->rho_mode; P.beta_mode = ctl->beta_mode;
    P.check_solved = ctl->check_solved < 1 ? 1 : ctl->check_solved;
    P.adaptive_rho = ctl->adaptive_rho;
    P.eps_abs = (T)ctl->eps_abs; P.eps_rel = (T)ctl->eps_rel; P.rho_value = (T)ctl->rho_value;
    P.rho_min = (T)ctl->rho_min; P.rho_max = (T)ctl->rho_max;
    P.ar_tol = (T)ctl->adaptive_rho_tol; P.ar_inv_tol = (T)(1.0 / ctl->adaptive_rho_tol);
    P.ar_thr = (T)ctl->adaptive_rho_threshold; P.beta_value = (T)ctl->beta_value;

    const int check = P.check_solved;
    const int max_iters = ctl->max_iters;
    const int ar_iter = ctl->adaptive_rho_iter < 1 ? 1 : ctl->adaptive_rho_iter;
    int n_launch = 0;

    // ---- x-update linear algebra: pivoted LU of the KKT matrix, or the symmetric inverse (lqp_spd.hpp) ----
    bool spd = false;
    if constexpr (sizeof(T) == 4) {
        int want = ctl->linsolve;
        if (want == 0) want = knobs().linsolve;
        const bool rho_pos = !(ctl->rho_mode == 1 && !(ctl->rho_value > 0.0));
        // (512 < n <= 1024: the sweep parks its panel in the M area, see wg_spd_sweep_big)
        spd = !force_lu && want != 1 && rho_pos && P.Ks <= SPD_BIGK && m <= SPD_MAXM &&
              (P.Ks <= SPD_MAXK || ((size_t)P.Np * P.Np >= (size_t)P.Ks * LQP_BLK && knobs().spd_big));
        // the equality correction (G, T: 2 m rows of 64 Ks floats next to the product's scratch) and the loop's vectors must
        // fit the 160 KB of LDS: large n with many equality rows (n > 960 at m >= 8, ...) stays on the LU path
        spd = spd && spd_factor_lds_bytes(m, P.Ks) <= 160 * 1024 && sym_loop_lds_bytes(n, m, P.Ks, 0) <= 160 * 1024;
    }
    P.spd = spd ? 1 : 0;
    // (the LU path too: its refactorisations form (D Q) D again from Q and the scaling vector -- assemble_kkt_rows -- instead of
    //  reading a stored copy the setup kernel would have to write: 128 MB per batch at n = 500)
    P.qs_lazy = knobs().qs_lazy ? 1 : 0;
    P.ar_iter = ar_iter; P.ar_max = ctl->adaptive_rho_max_iter; P.ring = kRing;

    // symmetric path with fewer problems than half the CUs: share each matrix between SPD_NP workgroups
    bool spd_split = false;
    const int spd_pivot_tasks = knobs().spd_ptasks;
    if (spd && P.Ks >= 3 && P.Ks <= SPD_MAXK) {
        int dev_ = 0, cus_ = 0;
        if (current_device_cus(&dev_, &cus_)) spd_split = B * SPD_NP <= cus_;
        spd_split = (knobs().spd_split < 0 ? (spd_split ? 1 : 0) : knobs().spd_split) != 0;
        spd_split = spd_split && (size_t)P.Np * P.Np >= 2 * 64 * SPD_LS;     // room for W, W^T in the M area
        spd_split = spd_split && !solo;
    }
    // More matrices than half the CUs (BASELINE configs[4]: 1024 per GPU): the resident sweep all the same, its pairs taking turns
    // on the chip as the loop's do (FwdParams::split_seg) -- partners are neighbours in dispatch order on one XCD (shared_map), so a
    // workgroup that waits for its partner waits for a CU that a COMPLETE earlier pair is about to leave.  Round 4 measured this
    // form slower than the one-workgroup sweep (8 turns x 0.36 ms against 2.9 ms); with the panel products on the float16 pipe a
    // turn is 0.28 ms and needs no pass over Q in front (k_spd_prep: 0.44 ms at B = 1024).
    bool spd_turns = false;
#if LQP_PIV_MFMA
    if (spd && !spd_split && !solo && P.xchg && P.Ks >= SPLIT_MINK && P.Ks <= SPD_MAXK && knobs().spd_turns != 0 &&
        knobs().spd_resident != 0 && knobs().spd_f16 != 0 && knobs().spd_split != 0 &&
        (size_t)P.Np * P.Np >= rs2_xb_floats(P.Ks) && (size_t)P.Np * P.Np >= 2 * 64 * SPD_LS) {
        int dev_ = 0, cus_ = 0;
        if (current_device_cus(&dev_, &cus_) && B * SPD_NP > cus_) { spd_turns = true; spd_split = true; }
    }
#endif
    bool spd_big_split = false;
    if (spd && P.Ks > SPD_MAXK) {
        int dev_ = 0, cus_ = 0;
        if (current_device_cus(&dev_, &cus_)) spd_big_split = B * SPD_NP <= cus_;
        spd_big_split = (knobs().spd_split < 0 ? (spd_big_split ? 1 : 0) : knobs().spd_split) != 0 && !solo;
    }
    // ... and, with the exchange buffer in the M area and the step flags behind the loop's granules, in ONE launch
    // (More matrices than half the CUs: the register-resident sweep with its pairs taking turns on the chip, as the loop does it
    //  -- FwdParams::split_seg -- was built and measured in round 4: the one-workgroup sweep streams every tile through L2 / HBM
    //  in every pivot step, 9.7 GB per factorisation at B = 1024, n = 500, and still is as fast: 2.91 ms against 8 turns x 0.36
    //  ms + the equality correction in a launch of its own = 3.13 ms; B = 256: 0.73 against 0.80.  Not kept.)
    bool spd_resident = spd_split && P.xchg && P.Ks >= SPLIT_MINK &&
                        (size_t)P.Np * P.Np >= rs2_xb_floats(P.Ks) &&
                        knobs().spd_resident != 0;
    int rs_np = SPD_NP;                 // workgroups per matrix of the resident sweep: 2, or 4 for batches up to a quarter of the CUs
    void (*rs_fn)(const FwdParams<float>, const int*) = nullptr;
    if (spd_resident) {
        // its workgroups per matrix wait for each other inside the launch: every one of them must be resident -- ask the
        // occupancy calculator for THIS kernel (block size, registers, LDS), not just the CU count
        int dev_ = 0, cus_ = 0, per_cu = 0;
        const int rlds = rs_q_lds_bytes(P.Ks);     // (the look-ahead sweep's counters behind the flags, the scaling vector behind them)
        bool ok = false;
#if LQP_PIV_MFMA
        if (P.Ks >= 7 && current_device_cus(&dev_, &cus_) && 4 * B <= cus_ && knobs().spd_resident4 != 0) {
            rs_fn = P.Ks == 7 ? k_spd_resident<7, 4> : k_spd_resident<8, 4>;
            if (knobs().spd_f16 != 0) rs_fn = P.Ks == 7 ? k_spd_resident<7, 4, true> : k_spd_resident<8, 4, true>;
            ok = ensure_lds((const void*)rs_fn, rlds) == LQP_OK && blocks_per_cu(&per_cu, rs_fn, RS_NT, rlds, dev_) &&
                 per_cu >= 1 && shared_grid(B, 4) <= cus_ * per_cu;
            if (ok) rs_np = 4;
        }
#endif
        if (!ok) {
            rs_fn = P.Ks == 3 ? k_spd_resident<3> : P.Ks == 4 ? k_spd_resident<4> : P.Ks == 5 ? k_spd_resident<5> : P.Ks == 6 ? k_spd_resident<6>
                  : P.Ks == 7 ? k_spd_resident<7> : k_spd_resident<8>;
#if LQP_PIV_MFMA
            if (knobs().spd_f16 != 0)
                rs_fn = P.Ks == 3 ? k_spd_resident<3, 2, true> : P.Ks == 4 ? k_spd_resident<4, 2, true> : P.Ks == 5 ? k_spd_resident<5, 2, true>
                      : P.Ks == 6 ? k_spd_resident<6, 2, true> : P.Ks == 7 ? k_spd_resident<7, 2, true> : k_spd_resident<8, 2, true>;
#endif
            ok = ensure_lds((const void*)rs_fn, rlds) == LQP_OK && current_device_cus(&dev_, &cus_) &&
                 blocks_per_cu(&per_cu, rs_fn, RS_NT, rlds, dev_) && per_cu >= 1 && (spd_turns || shared_grid(B, SPD_NP) <= cus_ * per_cu);
        }
        spd_resident = ok;
    }
    if (spd_turns && !spd_resident) { spd_turns = false; spd_split = false; }      // (the one-workgroup sweep after all)
    // rho = ||Qs||_F / sqrt(n): the norm is summed by k_spd_begin, which reads all of Q anyway, and rho is added to the
    // diagonal by the resident sweep -- the setup kernel then makes one pass over Q instead of two
    P.rho_late = (spd_resident && (!P.scale || P.qs_lazy) && ctl->rho_mode == 0 && knobs().rho_late) ? 1 : 0;
    // ... and above 512 rows on two workgroups per matrix: the sums k_spd_begin leaves are turned into rho by k_spd_rho_big
    const bool rho_late_big = spd_big_split && !spd_resident && (!P.scale || P.qs_lazy) && ctl->rho_mode == 0 && knobs().rho_late != 0 &&
                              (size_t)P.Np * P.Np >= (size_t)(2 * P.Ks + 1) * LQP_BLK;
    if (rho_late_big) P.rho_late = 1;
    // auto-scaling on: ONE pass over Q for the column maxima, the symmetry verdict and the (unscaled) blocks, in front of
    // the setup kernel (k_spd_prep); the resident sweep scales its tiles as it loads them and sums ||Qs||_F itself
    P.prep_fused = 0;
#if LQP_PIV_MFMA
    if constexpr (sizeof(T) == 4) {
        P.prep_fused = (spd_resident && P.scale && P.qs_lazy && (ctl->rho_mode != 0 || P.rho_late) && knobs().prep_fused) ? 1 : 0;
        // ... and the one-workgroup tier (more problems than half the CUs, n <= 512): k_spd_inverse scales the prepared
        // blocks, sums ||Qs||_F and adds rho itself (wg_spd_factor)
        if (!P.prep_fused && spd && !spd_split && !spd_big_split && P.Ks <= SPD_MAXK && P.scale && P.qs_lazy &&
            ctl->rho_mode != 2 && (size_t)n * P.ldq >= (size_t)SPD_NP * P.Ks * LQP_NB + 8 &&      // (k_spd_prep's scratch per
            knobs().prep_one)                                                             //  problem lives in its Qs area)
            P.prep_fused = 2;
    }
#endif
#if LQP_PIV_MFMA
    // ... or no pass in front at all: the resident sweep reads Q itself, straight into the registers that keep it
    if (P.prep_fused == 1 && (size_t)n * P.ldq >= (size_t)rs_np * (64 * P.Ks + 2) && knobs().qpass) P.prep_fused = 3;
#endif
    if constexpr (sizeof(T) == 4) {
        if (P.prep_fused == 1 || P.prep_fused == 2) {
            const int lds = (2 * 64 * SPD_LS + 2 * LQP_NW + 64 * P.Ks) * 4;
            const int r3 = ensure_lds((const void*)k_spd_prep<>, lds);
            if (r3) return r3;
            ProfScope ps(st, PC_SPD_INV);
            hipLaunchKernelGGL(k_spd_prep<>, dim3(shared_grid(B, SPD_NP)), dim3(LQP_NT), lds, st, P);
            ++n_launch;
        }
    }

    // ---- setup (its workgroup 0 zeroes status + counter ring), factor, pack ----
    {
        const int lds = setup_lds_bytes<T>(n);
        auto fn = k_fwd_setup<T>;
        int rc = ensure_lds((const void*)fn, lds);
        if (rc) return rc;
        ProfScope ps(st, PC_SETUP);
        hipLaunchKernelGGL(fn, dim3(B), dim3(LQP_NT), lds, st, P);
        ++n_launch;
    }
    // factorise (gate == nullptr) or refactorise under the device-side gate of k_rho_update
    auto factor_step = [&](const int* gate) -> int {
        if constexpr (sizeof(T) == 4) {
            if (spd) {
                const int lds = spd_factor_lds_bytes(m, P.Ks);
                auto inv_fn = P.Ks > SPD_MAXK ? k_spd_inverse<2> : k_spd_inverse<1>;
                const int r2 = ensure_lds((const void*)inv_fn, lds);
                if (r2) return r2;
                ProfScope ps(st, PC_SPD_INV);
                if (spd_big_split) {
                    // 512 < n <= 1024, few problems: two workgroups per matrix, a launch per phase of a pivot step
                    int r3 = ensure_lds((const void*)k_spd_begin<>, lds);
                    if (!r3) r3 = ensure_lds((const void*)k_spd_big_step<>, lds);
                    if (!r3) r3 = ensure_lds((const void*)k_spd_end<>, lds);
                    if (r3) return r3;
                    hipLaunchKernelGGL(k_spd_begin<>, dim3(shared_grid(B, SPD_NP)), dim3(LQP_NT), lds, st, P, gate);
                    if (rho_late_big && gate == nullptr) {
                        hipLaunchKernelGGL(k_spd_rho_big<>, dim3(B), dim3(LQP_NT), 0, st, P);
                        ++n_launch;
                    }
                    // two pivot steps per pass over the tiles (wg_spd_sweep_big, phases bits 2 .. 4): pivot + Y of step k | its update on
                    // block column k + 1 alone | pivot + Y' of step k + 1 | ONE read-modify-write of every other tile with both products
                    const bool fuse2 = knobs().spd_big_fuse != 0 && (size_t)P.Np * P.Np >= (size_t)(2 * P.Ks + 1) * LQP_BLK;
                    // ... its panel blocks as two-half operands, the update products on the float16 pipe (lqp_f16x2.hpp)
                    const int f16 = (fuse2 && knobs().spd_big_f16 != 0) ? 32 : 0;
                    for (int k = 0; k < P.Ks; ++k) {
                        const dim3 grd(shared_grid(B, SPD_NP)), blk(LQP_NT);
                        if (fuse2 && k + 1 < P.Ks) {
                            hipLaunchKernelGGL(k_spd_big_step<>, grd, blk, lds, st, P, gate, k, 1 | f16);
                            hipLaunchKernelGGL(k_spd_big_step<>, grd, blk, lds, st, P, gate, k, 4 | f16);
                            hipLaunchKernelGGL(k_spd_big_step<>, grd, blk, lds, st, P, gate, k + 1, 1 | 16 | f16);
                            hipLaunchKernelGGL(k_spd_big_step<>, grd, blk, lds, st, P, gate, k, 8 | f16);
                            ++k;
                            continue;
                        }
                        hipLaunchKernelGGL(k_spd_big_step<>, grd, blk, lds, st, P, gate, k, 1);
                        hipLaunchKernelGGL(k_spd_big_step<>, grd, blk, lds, st, P, gate, k, 2);
                    }
                    hipLaunchKernelGGL(k_spd_end<>, dim3(B), dim3(LQP_NT), lds, st, P, gate);
                    n_launch += 2 * P.Ks + 2;
                    return LQP_OK;
                }
                if (spd_split) {
                    // few problems: SPD_NP workgroups per matrix, one launch per pivot step (k_spd_begin/step/end)
                    int r3 = ensure_lds((const void*)k_spd_begin<>, lds);
                    if (!r3) r3 = ensure_lds((const void*)k_spd_step<>, lds);
                    if (!r3) r3 = ensure_lds((const void*)k_spd_end<>, lds);
                    if (r3) return r3;
                    if (!(gate == nullptr && P.prep_fused))      // (the first factorisation's blocks: k_spd_prep built them)
                        hipLaunchKernelGGL(k_spd_begin<>, dim3(shared_grid(B, SPD_NP)), dim3(LQP_NT), lds, st, P, gate);
                    if (spd_resident) {
                        // all pivot steps in one launch, the matrix in the registers of its two workgroups
                        const int rlds = rs_q_lds_bytes(P.Ks);
                        r3 = ensure_lds((const void*)rs_fn, rlds);
                        if (r3) return r3;
                        hipLaunchKernelGGL(rs_fn, dim3((knobs().dbg_loop_absent & 2) ? B : shared_grid(B, rs_np)), dim3(RS_NT), rlds, st, P, gate);
                        n_launch += 1;
                    } else {
                        for (int k = 0; k < P.Ks; ++k)
                            hipLaunchKernelGGL(k_spd_step<>, dim3(shared_grid(B, SPD_NP)), dim3(LQP_NT), lds, st, P, gate, k, spd_pivot_tasks);
                        n_launch += P.Ks;
                    }
                    if (!(gate == nullptr && P.eq_in_loop)) hipLaunchKernelGGL(k_spd_end<>, dim3(B), dim3(LQP_NT), lds, st, P, gate);
                    n_launch += 2;
                    return LQP_OK;
                }
                hipLaunchKernelGGL(inv_fn, dim3(B), dim3(LQP_NT), lds, st, P, gate);
                ++n_launch;
                return LQP_OK;
            }
        }
        // (scratch of the two-workgroup LU: the head of each problem's packed area -- the pack kernel fills it afterwards)
        const size_t pk_words = packed_blocks(P.K) * LQP_BLK * sizeof(T) / 8;
        int r2 = launch_lu(st, P.M, B, P.N, P.Np, (size_t)P.Np * P.Np, P.piv, P.Np, P.info, gate, nullptr,
                           (unsigned long long*)P.packed, pk_words);
        if (r2) return r2;
        r2 = launch_pack<T>(st, B, P.M, P.N, P.Np, (size_t)P.Np * P.Np, P.piv, P.Np, P.packed, P.dest, gate);
        n_launch += 2;
        return r2;
    };
    int rc = LQP_OK;

    // ---- launch mode ----
    // hot (first) launch: optional 512-thread build (256 VGPRs per thread: 16 register-resident blocks instead
    // of 8).  Measured SLOWER at B=128 n=500 (2.03 ms vs 1.54 ms; 1.76 ms with a 6-deep ring and 4 spills):
    // 8 waves hide the per-block dependent chain worse than 16, which costs more than the 8 extra resident
    // blocks save.  Kept selectable (LQP_LOOP512=1) for re-measurement; continuation launches always use 1024.
    const bool res_env = knobs().resident != 0;
    const bool resident = loop_resident_ok<1024>(P.K, sizeof(T)) && res_env &&
                          loop_lds_bytes<T>(n, m, P.Np, true) <= 160 * 1024;
    const bool hot512 = resident && loop_resident_ok<512>(P.K, sizeof(T)) && knobs().loop512 != 0;
    int loop_lds = loop_lds_bytes<T>(n, m, P.Np, resident);
    int loop_nt = hot512 ? 512 : 1024;
    auto loop_fn = k_admm_loop<T, false, false, 1024>;
    auto tail_fn = k_admm_loop<T, false, true, 1024>;        // same code, own name: continuation launches
    // the continuation kernel also runs LU + pack (in-kernel adaptive-rho refactor): LDS = max of the three
    int tail_lds = 0;
    // (above 1024 rows the LU kernel is the two-rows-per-thread one, launched on its own: refactorisations go through the
    //  separate gated kernels, and the continuation kernel's LDS does not have to hold an LU panel)
    // (float64 too since round 5: the pipelined schedule enqueues every possible event up front -- nine at the defaults -- and with
    //  separate gated kernels each one cost four launches that do nothing when the solve is over: rho update, LU, pack, tail)
    bool inkernel_refactor = P.N <= 1024;
    if constexpr (sizeof(T) == 4) {
        if (resident) { loop_fn = k_admm_loop<T, true, false, 1024>; tail_fn = k_admm_loop<T, true, true, 1024>; }
        if (hot512) loop_fn = k_admm_loop<T, true, false, 512>;
        if (spd) {
            P.sym_rl = sym_resident_lds_blocks(n, m, P.Ks);
            loop_lds = sym_loop_lds_bytes(n, m, P.Ks, P.sym_rl);
            tail_lds = std::max(loop_lds, spd_factor_lds_bytes(m, P.Ks));      // the tail refactorises in-kernel
            loop_nt = 1024;
            loop_fn = k_admm_loop<T, true, false, 1024, true>;
            tail_fn = k_admm_loop<T, true, true, 1024, true>;
            // optional 512-thread first launch (8 elements per thread, 12 register-resident blocks).  Measured at
            // B=128 n=500: 0.96-0.99 ms with an 8-deep ring (26 spilled VGPRs), 0.87 ms with a 6-deep one, against
            // 0.885 ms for the 1024-thread kernel: neither the halved instruction count nor the fewer streamed
            // blocks show, the product is bound by the un-overlapped sum of its resident and streamed phases.
            if (knobs().sym512) {
                P.sym_rl_hot = sym_resident_lds_blocks(n, m, P.Ks, resident_regs<512>(), 8);
                loop_lds = sym_loop_lds_bytes(n, m, P.Ks, P.sym_rl_hot, 8);
                loop_nt = 512;
                loop_fn = k_admm_loop<T, true, false, 512, true>;
            }
        }
    }
    rc = ensure_lds((const void*)loop_fn, loop_lds);
    if (rc) return rc;
    if (!spd) {
        tail_lds = loop_lds;
        if (P.N <= 1024) {
            tail_lds = std::max(tail_lds, LuLds<T, (sizeof(T) == 4 ? 16 : 8)>(round_up(P.N, 64)).total);
            tail_lds = std::max(tail_lds, (int)pack_lds_bytes<T>());
        }
    }
    rc = ensure_lds((const void*)tail_fn, tail_lds);
    if (rc) return rc;
    int mode = ctl->launch_mode;
    if (mode == 0) mode = knobs().launch_mode;     // auto: persistent when every workgroup is resident
    if (ctl->check_hook || solo) mode = 1;                   // the hook sits between the check segments | no grid barrier
    if (mode == 2) {
        // the grid barrier needs EVERY workgroup resident -- of the hot kernel and of the continuation kernel
        // (own block size and LDS footprint): take the smaller of the two answers
        int dev = 0, cus = 0, per_cu = 0, per_cu_tail = 0;
        if (!current_device_cus(&dev, &cus) || !blocks_per_cu(&per_cu, loop_fn, loop_nt, loop_lds, dev) ||
            !blocks_per_cu(&per_cu_tail, tail_fn, LQP_NT, tail_lds, dev))
            return LQP_ERR_HIP;
        per_cu = std::min(per_cu, per_cu_tail);
        if (per_cu < 1 || B > cus * per_cu) mode = 1;     // not every workgroup resident: no grid barrier
        // More problems than half the CUs but all resident (128 < B <= 256 here): the persistent one-workgroup kernel streams
        // two thirds of its matrix per iteration, while pairs of the two-workgroup kernel, taking turns, hold theirs in
        // registers (split_seg below) -- one launch per check segment then beats the persistent launch (B = 256, n = 500: loop
        // 0.82 -> 0.42 ms, step 1.87 -> 1.60 ms; B = 192: 1.63 -> 1.51).  Only when the caller left the mode to the library.
        if constexpr (sizeof(T) == 4) {
            if (mode == 2 && ctl->launch_mode == 0 && knobs().launch_mode == 2 && spd && P.xchg && 2 * B > cus &&
                P.Ks >= SPLIT_MINK && P.Ks <= SPD_MAXK && check >= 4 && knobs().loop_split != 0 &&
                knobs().loop_split_seg != 0 && !ctl->check_hook)
                mode = 1;
        }
    }
    const int max_checks_per_launch = kRing / 4;
    // two workgroups per QP for the first (hot) launch: symmetric path, persistent mode, 2 B workgroups resident
    bool loop_split = false;
    int split_lds = 0, split_nt = 512, loop_np = 1;
    void (*split_fn)(const FwdParams<float>, const int, const int, const int) = nullptr;
    if constexpr (sizeof(T) == 4) {
        if (spd && mode == 2 && P.xchg && P.Ks >= SPLIT_MINK && P.Ks <= SPD_MAXK && check >= 4 && knobs().loop_split != 0) {
            // 512 threads x 256 VGPRs: every block of the workgroup's half lives in registers.  (The 1024-thread
            // build -- 12 blocks in 128 VGPRs, 6 in LDS, 16 waves -- spills ~60 VGPRs into the hot loop and measured
            // 0.38 ms against 0.28 ms at B = 128, n = 500; the template still takes NT = 1024.)
            split_nt = 512;
            int dev = 0, cus = 0, per_cu = 0;
            if (!current_device_cus(&dev, &cus)) return LQP_ERR_HIP;
            // four workgroups per QP (one column pair each) when the batch leaves room for them: B <= #CUs / 4
            if (P.Ks >= 7 && 4 * B <= cus && knobs().loop_split4 != 0) {
                split_lds = split_loop_lds_bytes<512, 4>(P.Ks, m);
                split_fn = P.Ks == 7 ? k_admm_loop_split<7, 512, false, 4> : k_admm_loop_split<8, 512, false, 4>;
                if (split_lds <= 160 * 1024 && ensure_lds((const void*)split_fn, split_lds) == LQP_OK &&
                    blocks_per_cu(&per_cu, split_fn, split_nt, split_lds, dev) && per_cu >= 1 && shared_grid(B, 4) <= cus * per_cu)
                    loop_np = 4;
            }
            if (loop_np != 4) {
                split_lds = split_loop_lds_bytes<512>(P.Ks, m);
                split_fn = P.Ks == 3 ? k_admm_loop_split<3, 512> : P.Ks == 4 ? k_admm_loop_split<4, 512> : P.Ks == 5 ? k_admm_loop_split<5, 512> : P.Ks == 6 ? k_admm_loop_split<6, 512>
                         : P.Ks == 7 ? k_admm_loop_split<7, 512> : (g_lu_dbg ? k_admm_loop_split<8, 512, true> : k_admm_loop_split<8, 512>);
                if (split_lds <= 160 * 1024 && ensure_lds((const void*)split_fn, split_lds) == LQP_OK &&
                    blocks_per_cu(&per_cu, split_fn, split_nt, split_lds, dev) && per_cu >= 1 && shared_grid(B, 2) <= cus * per_cu)
                    loop_np = 2;
            }
            loop_split = loop_np > 1;
        }
    }
    // above 512 rows (BASELINE configs[3]: n = 1000): the streaming loop of the symmetric path with TWO workgroups per problem, each
    // streaming one range of whole block columns of H (admm_loop_body_from, NP == 2): twice the registers and LDS under the same
    // matrix (16 % -> 32 % of it on chip), half the stream per CU -- one CU alone pulls 58 GB/s, 29 us per iteration at n = 1000
    bool loop_np2 = false;
    if constexpr (sizeof(T) == 4) {
        if (spd && mode == 2 && P.xchg && P.Ks > SPD_MAXK && P.Ks <= SPD_BIGK && loop_nt == 1024 && knobs().loop_np2 != 0 && !solo &&
            !(retry & 2)) {
            int dev = 0, cus = 0, per_cu = 0;
            if (current_device_cus(&dev, &cus) && ensure_lds((const void*)k_admm_loop_np2<>, loop_lds) == LQP_OK &&
                blocks_per_cu(&per_cu, k_admm_loop_np2<>, 1024, loop_lds, dev) && per_cu >= 1 && shared_grid(B, 2) <= cus * per_cu)
                loop_np2 = true;
        }
    }
    // ... and for batches LARGER than half the CUs (BASELINE configs[4]: 1024 per GPU), where the loop runs one launch per
    // check segment anyway: the same kernel, its pairs taking turns on the chip (FwdParams::split_seg) -- a pair holds its
    // whole matrix in registers for the 15 iterations of a segment where the one-workgroup kernel streams two thirds of it
    // per iteration (B = 1024, n = 500: loop 3.35 -> 8 turns x 4 segments)
    bool loop_split_seg = false;
    if constexpr (sizeof(T) == 4) {
        if (spd && mode == 1 && !loop_split && P.xchg && P.Ks >= SPLIT_MINK && P.Ks <= SPD_MAXK && check >= 4 &&
            knobs().loop_split != 0 && knobs().loop_split_seg != 0 && !(retry & 2) && !solo) {
            int dev = 0, cus = 0, per_cu = 0;
            split_nt = 512;
            split_lds = split_loop_lds_bytes<512>(P.Ks, m);
            split_fn = P.Ks == 3 ? k_admm_loop_split<3, 512> : P.Ks == 4 ? k_admm_loop_split<4, 512> : P.Ks == 5 ? k_admm_loop_split<5, 512> : P.Ks == 6 ? k_admm_loop_split<6, 512>
                     : P.Ks == 7 ? k_admm_loop_split<7, 512> : k_admm_loop_split<8, 512>;
            if (current_device_cus(&dev, &cus) && 2 * B > cus && split_lds <= 160 * 1024 &&
                ensure_lds((const void*)split_fn, split_lds) == LQP_OK &&
                blocks_per_cu(&per_cu, split_fn, split_nt, split_lds, dev) && per_cu >= 1)
                loop_split_seg = true;
        }
    }
    // LU tier, persistent mode, n <= 256, 2 B workgroups resident: the explicit inverse of the KKT matrix in the registers of two
    // workgroups per problem (lqp_dense.hpp) -- float64, many equality rows, non-symmetric Q, control['linsolve'] = 'lu'
    bool loop_dense = false;
    int dense_lds = 0;
    if (!spd && mode == 2 && P.dnx && n <= DENSE_NMAX && knobs().loop_dense != 0) {
        int dev = 0, cus = 0, per_cu = 0;
        dense_lds = dense_loop_lds_bytes<T>(m);
        auto fnd = k_admm_loop_dense<T>;
        if (dense_lds <= 160 * 1024 && lu_inverse_lds_bytes<T>(P.Np) <= 160 * 1024 && current_device_cus(&dev, &cus) &&
            ensure_lds((const void*)fnd, dense_lds) == LQP_OK && blocks_per_cu(&per_cu, fnd, DENSE_NT, dense_lds, dev) &&
            per_cu >= 1 && shared_grid(B, 2) <= cus * per_cu)
            loop_dense = true;
    }
    // ... and for batches that leave most of the chip idle, any n the inverse kernel takes: W workgroups per problem, each with its
    // rows of the inverse in registers (k_admm_loop_dense_w): a matrix-vector product spreads over CUs, triangular solves do not
    bool loop_dense_w = false;
    int densew_lds = 0, densew_tprv = 0, densew_W = 0;
    if (!spd && mode == 2 && P.dnx && !loop_dense && n > DENSE_NMAX && knobs().loop_dense_w != 0 && lu_inverse_fits<T>(P.Np) &&
        (size_t)P.dnx_words >= densew_xchg_words<T>(n)) {
        int dev = 0, cus = 0, per_cu = 0;
        densew_lds = densew_lds_bytes<T>(n, m);
        densew_tprv = densew_tpr(n, densew_cpt<T>());
        densew_W = (n + DENSEW_NT / densew_tprv - 1) / (DENSEW_NT / densew_tprv);
        auto fnw = k_admm_loop_dense_w<T>;
        if (densew_tprv <= 32 && densew_W >= 2 && densew_lds <= 160 * 1024 && current_device_cus(&dev, &cus) &&
            ensure_lds((const void*)fnw, densew_lds) == LQP_OK && blocks_per_cu(&per_cu, fnw, DENSEW_NT, densew_lds, dev) &&
            per_cu >= 1 && densew_W * B <= cus * per_cu)
            loop_dense_w = true;
    }
    // small problems (n <= 128, e.g. BASELINE configs[1]): 256 threads per QP, the full matrix in registers (k_admm_loop_small)
    bool loop_small = false;
    int small_lds = 0;
    if constexpr (sizeof(T) == 4) {
        if (spd && mode == 2 && !loop_split && P.Ks <= 2 && knobs().loop_small != 0) {
            int dev = 0, cus = 0, per_cu = 0;
            small_lds = small_loop_lds_bytes(m);
            if (current_device_cus(&dev, &cus) && blocks_per_cu(&per_cu, k_admm_loop_small<>, 256, small_lds, dev) &&
                per_cu >= 1 && B <= cus * per_cu)
                loop_small = true;
        }
    }
    // the equality correction of the first factorisation moves into that kernel (its blocks are in registers there)
    // (ctrl.reserved2 bit 0: the caller will read the corrected H from the workspace afterwards -- lqp_boxqp_unroll_backward --
    //  so the correction must reach global memory: k_spd_end runs)
    if constexpr (sizeof(T) == 4)
    {
        // (one launch per check segment, pairs taking turns: the FIRST segment's launch applies it and writes the corrected blocks
        //  back -- from two turns of the chip on; below that k_spd_end on every CU at once is faster: B = 256 1.39 against 1.53 ms
        //  per step, B = 512 2.69 against 2.59)
        int dev_ = 0, cus_ = 0;
        const bool seg_eq = loop_split_seg && current_device_cus(&dev_, &cus_) && B >= 2 * cus_;
        P.eq_in_loop = ((loop_split || seg_eq) && spd_resident && m > 0 && !(ctl->reserved2 & 1) && knobs().eq_in_loop) ? 1 : 0;
    }
    rc = factor_step(nullptr);
    if (rc) return rc;
    if (loop_dense || loop_dense_w) {
        // X = M^-1 over the LAPACK copy of the factor (the pack kernel has read it; a refactorisation re-assembles M anyway)
        rc = launch_lu_inverse<T>(st, B, P.N, P.packed, packed_blocks(P.K) * LQP_BLK, P.dest, P.Np, P.M, (size_t)P.Np * P.Np, P.Np, nullptr);
        if (rc) return rc;
        ++n_launch;
    }
    // the first launch of the persistent modes
    auto launch_hot = [&](const int it, const int e, const int ctr_base, const int prev_slot, const int flags) {
        ProfScope ps(st, PC_LOOP);
        if constexpr (sizeof(T) == 4) {
            if (loop_split && it == 0) {
                hipLaunchKernelGGL(split_fn, dim3((knobs().dbg_loop_absent & 1) ? B : shared_grid(B, loop_np)), dim3(split_nt), split_lds, st, P, it, e, ctr_base);
                return;
            }
            if (loop_small && it == 0) {
                hipLaunchKernelGGL(k_admm_loop_small<>, dim3(B), dim3(256), small_lds, st, P, it, e, ctr_base);
                return;
            }
            if (loop_np2 && it == 0) {
                hipLaunchKernelGGL(k_admm_loop_np2<>, dim3((knobs().dbg_loop_absent & 8) ? B : shared_grid(B, 2)), dim3(1024), loop_lds, st, P, it, e, ctr_base, prev_slot, flags);
                return;
            }
        }
        if (loop_dense && it == 0) {
            hipLaunchKernelGGL(k_admm_loop_dense<T>, dim3(shared_grid(B, 2)), dim3(DENSE_NT), dense_lds, st, P, it, e, ctr_base);
            return;
        }
        if (loop_dense_w && it == 0) {
            hipLaunchKernelGGL(k_admm_loop_dense_w<T>, dim3(densew_W * B), dim3(DENSEW_NT), densew_lds, st, P, it, e, ctr_base, densew_tprv);
            return;
        }
        hipLaunchKernelGGL(loop_fn, dim3(B), dim3(loop_nt), loop_lds, st, P, it, e, ctr_base, prev_slot, flags);
    };

    // ---- no-host-sync plan (ctrl.reserved == 1, persistent launches): enqueue the WHOLE schedule.
    //      Every kernel exits at once when the device-side DONE flag is up, the refactor chains are
    //      gated on device, so nothing needs the host; errors are read later by the caller
    //      (lqp_boxqp_forward_layout).  Only taken when the number of adaptive-rho events is small.
    // A synchronous call (the default: the reference's semantics) enqueues the same schedule and then waits for the report
    // the last kernel stores into the caller's pinned memory (wait_report) -- not for the stream.
    if ((ctl->reserved >= 1 || knobs().sync_plan != 0) && mode == 2) {
        int n_events = 0;
        if (ctl->adaptive_rho)
            for (int a = ar_iter; a < max_iters && a < ctl->adaptive_rho_max_iter; a += ar_iter) ++n_events;
        if (n_events <= knobs().nosync_max_events) {
            int it = 0;
            const bool tail_epilogue = knobs().tail_epilogue != 0;
            bool epilogue_done = false;
            bool resume_next = false;       // the next continuation launch starts where the hot kernel says it stopped
            while (it < max_iters) {
                // the adaptive-rho step of iteration `it` runs as the prologue of the continuation kernel
                bool event = ctl->adaptive_rho && it > 0 && it % ar_iter == 0 && it < ctl->adaptive_rho_max_iter;
                if (event && !inkernel_refactor) {      // f64 / symmetric path: separate gated kernels
                    const int last_slot = ((it - 1) / check) % kRing;
                    { ProfScope ps(st, PC_RHO);
                      hipLaunchKernelGGL(k_rho_update<T>, dim3(B), dim3(LQP_NT), 0, st, P, last_slot, 0); }
                    ++n_launch;
                    rc = factor_step(P.status + ST_GATE);
                    if (rc) return rc;
                    event = false;
                }
                int e = max_iters;
                // (round 6) the two-workgroup loop does not stop at the first iteration at which rho MAY be adapted: it looks at the
                // counters of the check before and only hands over to the continuation kernel when the event changes something
                // (FwdParams::hot_past).  Its range is the whole solve; the continuation launch is enqueued from the first event
                // iteration on and finds its real starting point in status[ST_RESUME].
                const bool hot_past = it == 0 && loop_split && spd && inkernel_refactor && ctl->adaptive_rho && knobs().hot_past != 0 &&
                                      ar_iter < ctl->adaptive_rho_max_iter && ar_iter < max_iters;
                if (ctl->adaptive_rho && !((spd || inkernel_refactor) && it > 0) && !hot_past) {      // (the continuation kernel walks through its events itself)
                    const int a = (it / ar_iter + 1) * ar_iter;
                    if (a < ctl->adaptive_rho_max_iter) e = std::min(e, a);
                }
                // a launch may hold at most kRing/2 checks (counter ring); longer tails are split
                e = std::min(e, it + (kRing / 2) * check);
                const long long c_first = (it + check - 1) / check;
                const long long c_last = (e - 1) / check;
                if (c_last >= kRing) {      // these slots were used one ring ago: clear exactly them (stream-ordered)
                    for (long long c = std::max<long long>(c_first, kRing); c <= c_last; ) {
                        const int s0 = (int)(c % kRing);
                        const long long run = std::min<long long>(c_last - c + 1, kRing - s0);
                        HIP_OK(hipMemsetAsync(P.counters + (size_t)s0 * CT_WORDS, 0,
                                              sizeof(unsigned int) * CT_WORDS * (size_t)run, st));
                        c += run;
                    }
                }
                const int prev_slot = it > 0 ? ((it - 1) / check) % kRing : -1;
                bool last_tail = false;
                if (it == 0) {
                    P.hot_past = hot_past ? 1 : 0;
                    launch_hot(it, e, (int)(c_first % kRing), prev_slot, 1);
                    resume_next = hot_past;
                    if (hot_past) {
                        // A GIVEN rho is the case in which the adaptation does fire (`rho = 0.01`: three factorisations, 281 iterations,
                        // 4.5 ms with everything behind iteration 100 on the continuation kernel).  Enqueue a few rounds of {the event
                        // on the gated kernels of the refactorisation, the hot loop again from where it stopped}: each costs a solve that
                        // is over ~5 launches that leave at once -- which is why the automatic rho (it practically never adapts) gets none.
                        const int rounds = ctl->rho_mode != 0 ? knobs().hot_rounds : 0;
                        for (int rd = 0; rd < rounds; ++rd) {
                            { ProfScope ps(st, PC_RHO);
                              hipLaunchKernelGGL(k_rho_update<T>, dim3(B), dim3(LQP_NT), 0, st, P, -1, max_iters); }
                            ++n_launch;
                            rc = factor_step(P.status + ST_GATE);
                            if (rc) return rc;
                            if constexpr (sizeof(T) == 4) {
                                FwdParams<float> Pr = P;
                                Pr.hot_resume = 1;
                                Pr.eq_in_loop = 0;          // (k_spd_end has corrected the new blocks in global memory)
                                ProfScope ps(st, PC_LOOP);
                                hipLaunchKernelGGL(split_fn, dim3(shared_grid(B, loop_np)), dim3(split_nt), split_lds, st, Pr, ar_iter, e,
                                                   (int)(((ar_iter + check - 1) / check) % kRing));
                                ++n_launch;
                            }
                        }
                        e = ar_iter;          // (the continuation launch below is enqueued for [first event, ...): see above)
                    }
                } else {
                    // the last continuation launch ends with the epilogue (one launch and its boundary less)
                    last_tail = e >= max_iters && tail_epilogue;
                    ProfScope ps(st, PC_LOOP_TAIL);
                    hipLaunchKernelGGL(tail_fn, dim3(B), dim3(LQP_NT), tail_lds, st,
                                       P, it, e, (int)(c_first % kRing), prev_slot,
                                       ((event || spd || inkernel_refactor) ? 3 : 1) | (last_tail ? 4 : 0) | (resume_next ? 8 : 0));
                    resume_next = false;
                }
                ++n_launch;
                it = e;
                epilogue_done = last_tail;
            }
            if (!epilogue_done) {
                ProfScope ps(st, PC_EPILOGUE);
                hipLaunchKernelGGL(k_fwd_epilogue<T>, dim3(B), dim3(256), 0, st, P);
                ++n_launch;
            }
            if (hipGetLastError() != hipSuccess) return LQP_ERR_HIP;
            const int factor_launches = spd ? (spd_big_split ? 2 * P.Ks + 2 : spd_split ? (spd_resident ? 3 : P.Ks + 2) : 1) : 2;
            if (ctl->reserved == 1) {
                if (stats) {
                    memset(stats, 0, sizeof(*stats));
                    stats->iters = stats->n_factor = stats->n_solve = stats->n_check = -1;   // not known on the host
                    stats->fail_index = -1; stats->n_launch = n_launch; stats->mode_used = 3;
                    stats->any_lb = stats->any_ub = -1;
                    stats->linsolve_used = spd ? 2 : 1;
                    stats->factor_launches = factor_launches;
                    stats->loop_workgroups = loop_split ? loop_np : ((loop_dense || loop_np2) ? 2 : (loop_dense_w ? densew_W : 1));
                }
                return LQP_OK;
            }
            if (ctl->reserved == 2 && P.host_report) {
                // split call: everything is enqueued; the caller does its own host work (output views, bookkeeping) while
                // the GPU runs and then collects the report with lqp_boxqp_forward_finish
                if (stats) {
                    memset(stats, 0, sizeof(*stats));
                    stats->iters = stats->n_factor = stats->n_solve = stats->n_check = -1;
                    stats->fail_index = -1; stats->n_launch = n_launch; stats->mode_used = 4;
                    stats->any_lb = stats->any_ub = -1;
                    stats->linsolve_used = spd ? 2 : 1;
                    stats->factor_launches = factor_launches;
                    stats->loop_workgroups = loop_split ? loop_np : ((loop_dense || loop_np2) ? 2 : (loop_dense_w ? densew_W : 1));
                }
                return LQP_OK;
            }
            // ---- synchronous call: status block, info and flag words of every problem ----
            std::vector<int> rep_copy;
            const int* rep = P.host_report;
            if (!rep) {                     // (a C caller without pinned memory: three small copies and a stream wait)
                rep_copy.resize(ST_WORDS + 2 * B);
                HIP_OK(hipMemcpyAsync(rep_copy.data(), P.status, sizeof(int) * ST_WORDS, hipMemcpyDeviceToHost, st));
                HIP_OK(hipMemcpyAsync(rep_copy.data() + ST_WORDS, P.info, sizeof(int) * B, hipMemcpyDeviceToHost, st));
                HIP_OK(hipMemcpyAsync(rep_copy.data() + ST_WORDS + B, P.bflags, sizeof(int) * B, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                rep = rep_copy.data();
            }
            lqp_boxqp_stats local;
            lqp_boxqp_stats* so = stats ? stats : &local;
            memset(so, 0, sizeof(*so));
            so->n_launch = n_launch; so->linsolve_used = spd ? 2 : 1; so->factor_launches = factor_launches;
            so->loop_workgroups = loop_split ? loop_np : ((loop_dense || loop_np2) ? 2 : (loop_dense_w ? densew_W : 1));
            rc = collect_report(st, rep, rep == P.host_report, B, max_iters, check, so);
            if (rc == LQP_ERR_TIMEOUT && !spd && !t_single_wg_lu && !(flags_timeout_loop(rep))) {      // a shared LU timed out: one workgroup per matrix
                SingleWgLu only;
                return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats, ws, ws_bytes, retry);
            }
            if (rc == LQP_ERR_TIMEOUT && !solo)     // a shared sweep / loop / grid barrier gave up: once more with nothing shared
                return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats, ws, ws_bytes,
                                       retry | 4);
            if (rc == LQP_RETRY_LU)         // Qs + rho I not positive definite in f32 (first factorisation or an
                return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats,
                                       ws, ws_bytes, retry | 1);     // adaptive-rho one): the LU path takes the solve
            return rc;
        }
    }

    // ---- iterate ----
    int h_status[ST_WORDS];
    memset(h_status, 0, sizeof(h_status));
    int it = 0;
    long long zeroed_upto = kRing;     // check indices [0, zeroed_upto) have clean slots
    bool done = false, singular_checked = false;
    int nfactor_seen = 0;
    int chunk_cap = knobs().spec_launches;
    int fail_index = -1;
    while (it < max_iters && !done) {
        int in_chunk = 0;
        const int cap_now = (mode == 2) ? 1 : chunk_cap;       // persistent: the kernel itself stops at convergence
        while (it < max_iters && in_chunk < cap_now) {
            // adaptive-rho event at the start of iteration `it` (:237)
            const bool tail_events = spd && mode == 2;       // symmetric path, persistent: events inside the tail kernel
            if (!tail_events && ctl->adaptive_rho && it > 0 && it % ar_iter == 0 && it < ctl->adaptive_rho_max_iter) {
                const int last_slot = ((it - 1) / check) % kRing;
                { ProfScope ps(st, PC_RHO);
                  hipLaunchKernelGGL(k_rho_update<T>, dim3(B), dim3(LQP_NT), 0, st, P, last_slot, 0); }
                ++n_launch;
                rc = factor_step(P.status + ST_GATE);
                if (rc) return rc;
            }
            // end of this launch
            int e = max_iters;
            const int next_check = ((it + check - 1) / check) * check;
            if (mode == 1) e = std::min(e, next_check + 1);
            else e = std::min(e, next_check + 1 + (max_checks_per_launch - 1) * check);
            if (ctl->adaptive_rho && !(tail_events && it > 0)) {
                const int a = (it / ar_iter + 1) * ar_iter;
                if (a < ctl->adaptive_rho_max_iter) e = std::min(e, a);
            }
            if (e <= it) e = it + 1;
            // counter slots of the checks inside [it, e)
            const long long c_first = (it + check - 1) / check;        // first check index >= it
            const long long c_last = (e - 1) / check;                  // last check index <= e-1
            while (c_last >= zeroed_upto) {
                const int s0 = (int)(zeroed_upto % kRing);
                HIP_OK(hipMemsetAsync(P.counters + (size_t)s0 * CT_WORDS, 0, sizeof(unsigned int) * CT_WORDS * (kRing / 2), st));
                zeroed_upto += kRing / 2;
            }
            const int ctr_base = (int)(c_first % kRing);
            const int prev_slot = it > 0 ? ((it - 1) / check) % kRing : -1;
            { const bool first = (mode == 2 && it == 0) || mode == 1;
              if (first && mode == 2) {
                  launch_hot(it, e, ctr_base, prev_slot, 1);
              } else if (loop_split_seg) {
                  if constexpr (sizeof(T) == 4) {
                      FwdParams<float> Ps = P;
                      Ps.split_seg = 1;
                      Ps.seg_prev_slot = prev_slot;
                      if (it > 0) Ps.eq_in_loop = 0;
                      ProfScope ps(st, PC_LOOP);
                      hipLaunchKernelGGL(split_fn, dim3(shared_grid(B, 2)), dim3(split_nt), split_lds, st, Ps, it, e, ctr_base);
                  }
              } else {
                  ProfScope ps(st, first ? PC_LOOP : PC_LOOP_TAIL);
                  hipLaunchKernelGGL(first ? loop_fn : tail_fn, dim3(B), dim3(first ? loop_nt : LQP_NT), first ? loop_lds : tail_lds, st, P, it, e,
                                     ctr_base, prev_slot, mode == 2 ? ((tail_events && !first) ? 3 : 1) : 0);
              } }
            ++n_launch;
            ++in_chunk;
            it = e;
            if (ctl->check_hook && mode == 1 && ((e - 1) % check) == 0) {
                // strict global stop: all-reduce this check's counters over the ranks before anything reads them
                const long long c_idx = (e - 1) / check;
                if (ctl->check_hook(ctl->check_hook_user, (void*)st, (void*)(P.counters + (size_t)(c_idx % kRing) * CT_WORDS),
                                    (int)c_idx) != 0)
                    return LQP_ERR_HIP;
            }
        }
        // ---- close the chunk: did the last check stop the loop? ----
        if (it > 0 && ((it - 1) % check) == 0) {
            // did the last check of this chunk stop the loop?  (tiny kernel, own name in traces)
            const int prev_slot = ((it - 1) / check) % kRing;
            ProfScope ps(st, PC_MISC);
            hipLaunchKernelGGL(k_check_done<>, dim3(1), dim3(64), 0, st, P.status, P.counters, prev_slot, it - 1);
            ++n_launch;
        }
        // the epilogue only reads state: run it now so that the common case (converged in this chunk)
        // pays no host round trip between the loop and its outputs; a later chunk simply re-runs it
        { ProfScope ps(st, PC_EPILOGUE);
          hipLaunchKernelGGL(k_fwd_epilogue<T>, dim3(B), dim3(256), 0, st, P); }
        ++n_launch;
        // (with a host report the epilogue has left status and info words in pinned host memory: one wait, no copies)
        if (!P.host_report) HIP_OK(hipMemcpyAsync(h_status, P.status, sizeof(h_status), hipMemcpyDeviceToHost, st));
        auto fetch_status = [&]() -> int {
            HIP_OK(hipStreamSynchronize(st));
            if (P.host_report)
                for (int i = 0; i < ST_WORDS; ++i) h_status[i] = ((const volatile int*)P.host_report)[i];
            return LQP_OK;
        };
        // A failed factorisation ends or restarts the solve.  With a check hook (strict global stop over batch shards)
        // that decision must be the same on every rank -- a rank that restarted alone would pair its collectives with
        // other checks of its peers -- so the local verdict goes through the hook (SUM over the ranks) first.
        // returns LQP_OK, an error, or -1: repeat the solve on the LU path
        auto after_factorisation = [&]() -> int {
            int rcf = first_failure(st, P.info, B, &fail_index, P.host_report ? P.host_report + ST_WORDS : nullptr);      // synchronises
            if (rcf == LQP_OK || rcf == LQP_ERR_SINGULAR) { const int r4 = fetch_status(); if (r4) return r4; }
            if (rcf == LQP_ERR_TIMEOUT && !spd && !t_single_wg_lu) return -2;
            if (rcf != LQP_OK && rcf != LQP_ERR_SINGULAR) return rcf;
            bool leave_spd = rcf == LQP_ERR_SINGULAR && spd, singular = rcf == LQP_ERR_SINGULAR && !spd;
            if (ctl->check_hook) {
                int vote[CT_WORDS] = {leave_spd ? 1 : 0, singular ? 1 : 0, 0, 0};
                HIP_OK(hipMemcpyAsync(P.status + ST_VOTE, vote, sizeof(vote), hipMemcpyHostToDevice, st));
                if (ctl->check_hook(ctl->check_hook_user, (void*)st, (void*)(P.status + ST_VOTE), -1) != 0) return LQP_ERR_HIP;
                HIP_OK(hipMemcpyAsync(vote, P.status + ST_VOTE, sizeof(vote), hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                leave_spd = vote[0] > 0;
                singular = vote[1] > 0;
            }
            if (singular) {                                           // (fail_index -1: the singular problem is on another rank)
                if (stats) { memset(stats, 0, sizeof(*stats)); stats->fail_index = fail_index; }
                return LQP_ERR_SINGULAR;
            }
            return leave_spd ? -1 : LQP_OK;
        };
        if (!singular_checked) {
            singular_checked = true;
            rc = after_factorisation();
            if (rc == -1)       // Qs + rho I not positive definite in f32 (on some rank): the LU path takes it
                return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats,
                                       ws, ws_bytes, retry | 1);
            if (rc == -2) {     // a shared LU timed out: once more, one workgroup per matrix
                SingleWgLu only;
                return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats, ws, ws_bytes, retry);
            }
            if (rc) return rc;
        } else {
            { const int r4 = fetch_status(); if (r4) return r4; }
            if (h_status[ST_NFACTOR] != nfactor_seen) {      // an adaptive-rho refactorisation ran in this chunk
                rc = after_factorisation();
                if (rc == -1)
                    return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats,
                                           ws, ws_bytes, retry | 1);
                if (rc == -2) {
                    SingleWgLu only;
                    return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats, ws, ws_bytes, retry);
                }
                if (rc) return rc;
            }
        }
        nfactor_seen = h_status[ST_NFACTOR];
        if (h_status[ST_TIMEOUT]) {
            // The pairs of the turn-taking loop become resident together only while the dispatcher hands workgroups out in
            // blockIdx order onto an otherwise idle chip; when something else holds CUs (another stream, RCCL kernels) a pair can
            // wait for a partner that is not resident.  Its spin is bounded (0.5 s), the kernels drain, and the solve is run again
            // from its setup on the loop that needs nobody (degrade, do not fail).
            if (loop_split_seg && !(retry & 2))
                return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats, ws, ws_bytes,
                                       retry | 2);
            if (!solo)                      // (the same for every other shared schedule: retry bit 2)
                return forward_impl<T>(st, B, n, m, Q, p, A, b, lb, ub, ctl, rho_in, x, z, u, lams, nus, rho_out, stats, ws, ws_bytes,
                                       retry | 4);
            return LQP_ERR_TIMEOUT;
        }
        done = h_status[ST_DONE] != 0;
        chunk_cap = std::min(chunk_cap * 2, 64);
    }
    const int final_iter = done ? h_status[ST_FINAL_ITER] : max_iters - 1;

    if (hipGetLastError() != hipSuccess) return LQP_ERR_HIP;
    if (stats) {
        stats->iters = final_iter;
        stats->n_factor = 1 + h_status[ST_NFACTOR];
        stats->n_solve = final_iter + 1;
        stats->n_check = final_iter / check + 1;
        stats->rho_updated = h_status[ST_RHO_UPDATED];
        stats->fail_index = -1;
        stats->n_launch = n_launch;
        stats->mode_used = mode;
        stats->linsolve_used = spd ? 2 : 1;
        stats->factor_launches = spd ? (spd_big_split ? 2 * P.Ks + 2 : spd_split ? (spd_resident ? 3 : P.Ks + 2) : 1) : 2;
        stats->loop_workgroups = (loop_split && mode == 2) ? loop_np : (loop_split_seg || (loop_np2 && mode == 2)) ? 2 : 1;
        stats->any_lb = h_status[ST_ANY_LB]; stats->any_ub = h_status[ST_ANY_UB];
    }
    return LQP_OK;
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
template <typename T>
size_t carve_backward(void* ws, int B, int n, int m, BwdParams<T>& P) {
    P.B = B; P.n = n; P.m = m; P.N = n + m;
    P.Np = round_up(P.N, LQP_NB); P.K = P.Np / LQP_NB;
    Carver c(ws);
    P.info = c.take<int>(B);
    P.piv = c.take<int>((size_t)B * P.Np);
    P.dest = c.take<int>((size_t)B * P.Np);
    P.rhs = c.take<T>((size_t)B * P.Np);
    P.rhs2 = c.take<T>((size_t)B * P.Np);
    P.bsc = c.take<T>((size_t)B * P.Np);
    P.fidx = c.take<int>((size_t)B * n);
    P.nred = c.take<int>(B);
    P.M = c.take<T>((size_t)B * P.Np * P.Np);
    P.packed = c.take<T>((size_t)B * packed_blocks(P.K) * LQP_BLK);
    return c.off + kAlign;
}

template <typename T>
int backward_impl(hipStream_t st, int B, int n, int m, const void* g, const void* x, const void* u, const void* lams,
                  const void* nus, const void* Q, const void* A, const void* lb, const void* ub, int rho_mode,
                  double rho_value, const void* rho_in, void* dQ, void* dp, void* dA, void* db, void* dlb, void* dub,
                  int32_t* fail_index, void* ws, size_t ws_bytes, int linsolve, void* host_report, const int kkt = 0,
                  int phase = 0) {
    BwdParams<T> P;
    memset(&P, 0, sizeof(P));
    P.kkt = kkt ? 1 : 0;
    const bool reported = (linsolve & LQP_BWD_REPORTED) != 0;
    linsolve &= ~LQP_BWD_REPORTED;
    if (linsolve & LQP_BWD_PREFACTORED) { linsolve &= ~LQP_BWD_PREFACTORED; if (phase == 0 && !kkt) phase = 2; }
    P.host_report = (int*)host_report;
    // (the words are set to -1 before the first launch, below -- unless the prefactor call has reported into them and this call
    //  solves on its factor)
    bool reset_pending = P.host_report != nullptr;
    P.early_report = knobs().bwd_early != 0 ? 1 : 0;
    const size_t need = carve_backward<T>(ws, B, n, m, P);
    if (ws_bytes < need) return LQP_ERR_WORKSPACE;
    if (!knobs().bwd_equil) P.bsc = nullptr;
    P.g = (const T*)g; P.x = (const T*)x; P.u = (const T*)u; P.lams = (const T*)lams; P.nus = (const T*)nus;
    P.Q = (const T*)Q; P.A = (const T*)A; P.lb = (const T*)lb; P.ub = (const T*)ub; P.rho_in = (const T*)rho_in;
    P.rho_value = (T)rho_value; P.rho_mode = rho_mode;
    P.dQ = (T*)dQ; P.dp = (T*)dp; P.dA = (T*)dA; P.db = (T*)db; P.dlb = (T*)dlb; P.dub = (T*)dub;
    P.dbg = g_lu_dbg;
    // default: solve on the free set only (see k_bwd_build_reduced); LQP_BWD_FULL=1 keeps the full system
    P.reduced = (knobs().bwd_full && !kkt) ? 0 : 1;
    const int* nvec = P.reduced ? P.nred : nullptr;
    // linsolve 2: the caller vouches for a symmetric Q (the forward's symmetric x-update checked it): the
    // reduced system goes through a blocked Cholesky of Q_FF instead of the pivoted LU of the bordered matrix
    bool chol = false;
    if constexpr (sizeof(T) == 4) {
        chol = P.reduced && linsolve == 2 && knobs().bwd_chol && round_up(n, LQP_NB) / LQP_NB <= SPD_BIGK &&
               m <= SPD_MAXM && bwd_chol_lds_bytes(n, m, m >= 3 ? 4 : 2) <= 160 * 1024;
    }
    // the LU form has the same two phases (ABI 11): 1 = free set, reduced system, its pivoted LU and the packed factor -- none of
    // them needs the cotangent --, 2 = gather the cotangent over the free set, solve (+ refinement), epilogue
    const bool lu_phases = !chol && P.reduced && !kkt;
    if (!chol && !lu_phases) {
        if (phase == 1) return LQP_ERR_UNSUPPORTED;               // (the full-system form, the KKT backward: one call)
        phase = 0;
    }
    P.phase = phase;
    // (with LQP_BWD_EARLY=0 the epilogue reports: the words are this call's to reset and to wait for)
    P.reported = (reported && phase == 2 && P.host_report && P.early_report) ? 1 : 0;
    if (reset_pending && !P.reported) report_reset(P.host_report, B);
    reset_pending = false;
    if constexpr (sizeof(T) == 4) {
        if (chol) {
            P.chol = 1;
            if (phase != 2) {
                const int lds = (2 * round_up(n, 8) + LQP_NW + 8 + round_up(n, 64) + 64) * 4;      // (fl | wtot | the KKT form's diagonal weights | the equilibration)
                ProfScope ps(st, PC_BWD_BUILD);
                const int split = B <= 128 ? 2 : 1;
                hipLaunchKernelGGL(k_bwd_build_chol<>, dim3(B, split), dim3(LQP_NT), lds, st, P);
            }
            const bool nr4 = m >= 3;                   // (four right-hand sides per round of the block solves)
            const int lds = bwd_chol_lds_bytes(n, m, nr4 ? 4 : 2);
            {
                const int Kmax = round_up(n, LQP_NB) / LQP_NB;
                P.la_maxk = !knobs().bwd_lookahead ? 0 : (Kmax < SPD_MAXK ? Kmax : SPD_MAXK - 1);
            }
            auto chol_fn = nr4 ? k_bwd_chol_solve<4> : k_bwd_chol_solve<0>;
#if LQP_PIV_MFMA
            if (knobs().spd_f16 != 0 && knobs().bwd_f16 != 0) chol_fn = nr4 ? k_bwd_chol_solve<4, true> : k_bwd_chol_solve<0, true>;
#endif
            int r2 = ensure_lds((const void*)chol_fn, lds);
            if (r2) return r2;
            ProfScope ps(st, PC_BWD_CHOL);
            hipLaunchKernelGGL(chol_fn, dim3(B), dim3(LQP_NT), lds, st, P);
            if (phase == 1) return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
        }
    }
    if (chol) {
    } else if (phase == 2) {
        ProfScope ps(st, PC_BWD_BUILD);
        hipLaunchKernelGGL(k_bwd_gather_rhs<T>, dim3(B), dim3(256), 0, st, P);
    } else if (P.reduced) {
        const int lds = (round_up(n, 8) + LQP_NW + 8) * 4;
        ProfScope ps(st, PC_BWD_BUILD);
        const int split = (B <= 128 && knobs().split2) ? 2 : 1;
        hipLaunchKernelGGL(k_bwd_build_reduced<T>, dim3(B, split), dim3(LQP_NT), lds, st, P);
    } else {
        ProfScope ps(st, PC_BWD_BUILD);
        hipLaunchKernelGGL(k_bwd_build<T>, dim3(B), dim3(LQP_NT), 0, st, P);
    }
    int rc = LQP_OK;
    if (!chol) {
        if (phase != 2) {
            rc = launch_lu(st, P.M, B, P.N, P.Np, (size_t)P.Np * P.Np, P.piv, P.Np, P.info, nullptr, nvec,
                           (unsigned long long*)P.packed, packed_blocks(P.K) * LQP_BLK * sizeof(T) / 8);
            if (rc) return rc;
            // a caller that waits (or a prefactor call: its report buffer is the backward call's): the LU is the step that can
            // fail or time out -- its info words now, see k_report_info
            if (P.host_report && (phase == 1 || (P.early_report && fail_index))) {
                ProfScope ps(st, PC_MISC);
                hipLaunchKernelGGL(k_report_info<>, dim3((B + 255) / 256), dim3(256), 0, st, (const int*)P.info, P.host_report, B);
                P.lu_reported = 1;
            }
            rc = launch_pack<T>(st, B, P.M, P.N, P.Np, (size_t)P.Np * P.Np, P.piv, P.Np, P.packed, P.dest, nullptr, nvec);
            if (rc) return rc;
            if (phase == 1) return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
        } else if (P.reported) P.lu_reported = 1;
        rc = launch_solve<T>(st, B, P.packed, P.N, P.dest, P.rhs, 1, (size_t)P.Np, 1, 0, nvec);
        if (rc) return rc;
        if (P.reduced && !kkt && knobs().bwd_refine) {
            // one refinement step: residual with the original entries (double accumulation), correction solve
            const int lds = (round_up(n, 8) + round_up(m > 0 ? m : 1, 8)) * (int)sizeof(T) + round_up(n, 8) * 4;
            { ProfScope ps(st, PC_BWD_BUILD);
              hipLaunchKernelGGL(k_bwd_residual<T>, dim3(B, (B <= 128 && knobs().split2) ? 2 : 1), dim3(LQP_NT), lds, st, P); }
            rc = launch_solve<T>(st, B, P.packed, P.N, P.dest, P.rhs2, 1, (size_t)P.Np, 1, 0, nvec);
            if (rc) return rc;
            P.refine = 1;
        }
    }
    {
        const int lds = (2 * n + m + 8) * (int)sizeof(T);
        auto fn = k_bwd_epilogue<T>;
        rc = ensure_lds((const void*)fn, lds);
        if (rc) return rc;
        ProfScope ps(st, PC_BWD_EPILOGUE);
        const int slabs = (knobs().epi_slabs < 0 ? (B <= 128 ? 2 : 1) : knobs().epi_slabs);      // row slabs per problem: fill the chip when the batch is small
        hipLaunchKernelGGL(fn, dim3(B, slabs), dim3(LQP_NT), lds, st, P);
    }
    if (hipGetLastError() != hipSuccess) return LQP_ERR_HIP;
    if (fail_index) {
        int fi = -1;
        // torch.linalg.solve checks info (and waits) too.  The epilogue stores the info words into the caller's pinned
        // memory as it STARTS: the call returns while the gradients are still being written (stream-ordered results)
        if (P.host_report && knobs().sync_plan != 0) {
            rc = wait_report(st, P.host_report, B);
            if (rc) return rc;
            bool gave_up = false;                    // (-7: a hand-off of a shared LU timed out)
            for (int i = 0; i < B; ++i) {
                const int v = ((const volatile int*)P.host_report)[i];
                if (v != 0 && fi < 0) fi = i;
                gave_up = gave_up || v == -7;
            }
            rc = gave_up ? LQP_ERR_TIMEOUT : fi >= 0 ? LQP_ERR_SINGULAR : LQP_OK;
        } else
            rc = first_failure(st, P.info, B, &fi, P.host_report);
        if (rc == LQP_ERR_TIMEOUT && !chol && !t_single_wg_lu) {      // a shared LU timed out: once more, one workgroup per matrix
            SingleWgLu only;
            return backward_impl<T>(st, B, n, m, g, x, u, lams, nus, Q, A, lb, ub, rho_mode, rho_value, rho_in, dQ, dp, dA,
                                    db, dlb, dub, fail_index, ws, ws_bytes, 1, host_report, kkt, 0);
        }
        if (rc == LQP_ERR_SINGULAR && chol)           // Q_FF not positive definite in f32: the pivoted LU takes it
            return backward_impl<T>(st, B, n, m, g, x, u, lams, nus, Q, A, lb, ub, rho_mode, rho_value, rho_in, dQ, dp, dA,
                                    db, dlb, dub, fail_index, ws, ws_bytes, 1, host_report, kkt, 0);
        *fail_index = fi;
        if (rc) return rc;
    }
    return LQP_OK;
}

// ---------------------------------------------------------------------------
// LU / solve / KKT entry points
// ---------------------------------------------------------------------------
inline size_t lu_scratch_u64(int N) { return N > 512 ? (luw_scratch_words<double>(N) + 1) / 2 : (size_t)LU2_SCR_WORDS; }
template <typename T>
size_t carve_lu(void* ws, int B, int N, T*& M, int*& piv, unsigned long long*& scr) {
    const int Np = round_up(N, LQP_NB);
    Carver c(ws);
    piv = c.take<int>((size_t)B * Np);
    M = c.take<T>((size_t)B * Np * Np);
    scr = c.take<unsigned long long>((size_t)B * lu_scratch_u64(N));      // (hand-off words of the two-workgroup LU / messages of the wide one)
    return c.off + kAlign;
}

template <typename T>
int lu_factor_impl(hipStream_t st, int B, int N, void* Mio, int32_t* piv_out, int32_t* info_out, void* ws, size_t ws_bytes) {
    T* M; int* piv; unsigned long long* scr;
    const size_t need = carve_lu<T>(ws, B, N, M, piv, scr);
    if (ws_bytes < need) return LQP_ERR_WORKSPACE;
    const int Np = round_up(N, LQP_NB);
    hipLaunchKernelGGL(k_copy_matrix<T>, dim3(B), dim3(LQP_NT), 0, st, (const T*)Mio, N, (size_t)N * N, M, Np, (size_t)Np * Np, N);
    int rc = launch_lu(st, M, B, N, Np, (size_t)Np * Np, piv, Np, info_out, nullptr, nullptr, scr, lu_scratch_u64(N));
    if (rc) return rc;
    hipLaunchKernelGGL(k_copy_matrix<T>, dim3(B), dim3(LQP_NT), 0, st, (const T*)M, Np, (size_t)Np * Np, (T*)Mio, N, (size_t)N * N, N);
    hipLaunchKernelGGL(k_copy_ints<int>, dim3(B), dim3(256), 0, st, (const int*)piv, Np, (int*)piv_out, N, N);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

template <typename T> size_t packed_bytes_t(int B, int N) {
    const int K = round_up(N, LQP_NB) / LQP_NB;
    Carver c(nullptr);
    c.take<int>((size_t)B * K * LQP_NB);
    c.take<T>((size_t)B * packed_blocks(K) * LQP_BLK);
    return c.off + kAlign;
}
template <typename T> void carve_packed(void* buf, int B, int N, int*& dest, T*& packed) {
    const int K = round_up(N, LQP_NB) / LQP_NB;
    Carver c(buf);
    dest = c.take<int>((size_t)B * K * LQP_NB);
    packed = c.take<T>((size_t)B * packed_blocks(K) * LQP_BLK);
}

template <typename T>
int lu_pack_impl(hipStream_t st, int B, int N, const void* LU, const int32_t* piv, void* buf) {
    int* dest; T* packed;
    carve_packed<T>(buf, B, N, dest, packed);
    return launch_pack<T>(st, B, (const T*)LU, N, N, (size_t)N * N, (const int*)piv, N, packed, dest, nullptr);
}
template <typename T>
int lu_solve_packed_impl(hipStream_t st, int B, int N, int k, const void* buf, void* rhs) {
    int* dest; T* packed;
    carve_packed<T>((void*)buf, B, N, dest, packed);
    return launch_solve<T>(st, B, packed, N, dest, (T*)rhs, k, (size_t)N * k, k, 1);
}

template <typename T>
size_t carve_kkt(void* ws, int B, int n, int m, T*& M, T*& packed, T*& rhs, int*& piv, int*& dest, int*& info) {
    const int N = n + m, Np = round_up(N, LQP_NB), K = Np / LQP_NB;
    Carver c(ws);
    info = c.take<int>(B);
    piv = c.take<int>((size_t)B * Np);
    dest = c.take<int>((size_t)B * Np);
    rhs = c.take<T>((size_t)B * Np);
    M = c.take<T>((size_t)B * Np * Np);
    packed = c.take<T>((size_t)B * packed_blocks(K) * LQP_BLK);
    return c.off + kAlign;
}

template <typename T>
int kkt_solve_impl(hipStream_t st, int B, int n, int m, const void* Q, const void* p, const void* A, const void* b,
                   void* x, void* nus, int32_t* fail_index, void* ws, size_t ws_bytes) {
    T *M, *packed, *rhs; int *piv, *dest, *info;
    const size_t need = carve_kkt<T>(ws, B, n, m, M, packed, rhs, piv, dest, info);
    if (ws_bytes < need) return LQP_ERR_WORKSPACE;
    const int N = n + m, Np = round_up(N, LQP_NB);
    hipLaunchKernelGGL(k_kkt_build<T>, dim3(B), dim3(LQP_NT), 0, st, (const T*)Q, (const T*)p, (const T*)A, (const T*)b,
                       n, m, Np, M, rhs, info);
    int rc = launch_lu(st, M, B, N, Np, (size_t)Np * Np, piv, Np, info, nullptr, nullptr, (unsigned long long*)packed,
                       packed_blocks(Np / LQP_NB) * LQP_BLK * sizeof(T) / 8);
    if (rc) return rc;
    rc = launch_pack<T>(st, B, M, N, Np, (size_t)Np * Np, piv, Np, packed, dest, nullptr);
    if (rc) return rc;
    rc = launch_solve<T>(st, B, packed, N, dest, rhs, 1, (size_t)Np, 1, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_kkt_unpack<T>, dim3(B), dim3(256), 0, st, (const T*)rhs, n, m, Np, (T*)x, (T*)nus);
    if (hipGetLastError() != hipSuccess) return LQP_ERR_HIP;
    if (fail_index) {
        int fi = -1;
        rc = first_failure(st, info, B, &fi);
        *fail_index = fi;
        if (rc) return rc;
    }
    return LQP_OK;
}

bool bad_dims(int dtype, int B, int n, int m) {
    return (dtype != LQP_F32 && dtype != LQP_F64) || B < 1 || n < 1 || m < 0;
}

}  // namespace

// ===========================================================================
namespace {
// ---- the tape whose x-update is the pivoted LU (float32 / float64, any m): lqp_unroll.hpp, k_unroll_sweep_lu ----
template <typename T> struct UnrollLuCarve { UnrollLuParams<T> U; size_t bytes; };
template <typename T>
static UnrollLuCarve<T> carve_unroll_lu(void* ws, int B, int n, int m, int TT, bool segments = false) {
    UnrollLuCarve<T> c;
    memset(&c.U, 0, sizeof(c.U));
    Carver cv(ws);
    const int mm = m > 0 ? m : 1;
    c.U.X = cv.take<T>((size_t)B * TT * n);
    c.U.W = cv.take<T>((size_t)B * TT * n);
    c.U.DX = cv.take<T>((size_t)B * TT * n);
    c.U.NU = cv.take<T>((size_t)B * TT * mm);
    c.U.DNU = cv.take<T>((size_t)B * TT * mm);
    c.U.MK = cv.take<signed char>((size_t)B * TT * n);
    if (segments) {             // (a tape in segments also keeps z_{k+1}, u_{k+1}: the rho adaptation reads them)
        c.U.Zr = cv.take<T>((size_t)B * TT * n);
        c.U.Ur = cv.take<T>((size_t)B * TT * n);
    }
    c.U.inj_k = -1;
    c.bytes = cv.off + kAlign;
    return c;
}

// one segment [k0, k1) of a tape whose factor changes along it (a solve in which rho was adapted): replay and / or reverse walk
// with the epoch's packed factor and rho (lqp_unroll.hpp, UnrollLuParams)
template <typename T>
static int unroll_tape_segment_impl(hipStream_t st, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes, int iters,
                                    int k0, int k1, int mode, const void* packed_buf, const void* rho, void* state, int inj_k,
                                    const void* inj, const void* dl_dx, void* dps, void* dlbs, void* dubs, void* drho, void* dD,
                                    void* scratch, size_t scratch_bytes, void** zrows, void** urows, void** xrows) {
    FwdLayout<T> L = carve_forward<T>((void*)fwd_workspace, B, n, m);
    if (fwd_workspace_bytes < L.bytes) return LQP_ERR_WORKSPACE;
    const FwdParams<T>& P = L.P;
    const int TT = iters + 1;
    UnrollLuCarve<T> c = carve_unroll_lu<T>(scratch, B, n, m, TT, true);
    if (scratch_bytes < c.bytes) return LQP_ERR_WORKSPACE;
    if (zrows) *zrows = c.U.Zr;
    if (urows) *urows = c.U.Ur;
    if (xrows) *xrows = c.U.X;
    if (k1 <= k0) return LQP_OK;             // (a query of the row pointers)
    UnrollLuParams<T>& U = c.U;
    U.T_ = TT; U.k0 = k0; U.k1 = k1; U.mode = mode;
    if (packed_buf) {
        int* dest; T* packed;
        carve_packed<T>((void*)packed_buf, B, P.N, dest, packed);
        U.packed_ov = packed; U.dest_ov = dest;
    }
    U.rho_ov = (const T*)rho;
    U.state = (T*)state;
    U.inj_k = inj ? inj_k : -1; U.inj = (const T*)inj;
    U.g = (const T*)dl_dx;
    U.dps = (T*)dps; U.dlbs = (T*)dlbs; U.dubs = (T*)dubs; U.dD = (T*)dD; U.drho = (T*)drho;
    const int lds = unroll_lu_lds_bytes<T>(P.Np);
    auto fn = k_unroll_sweep_lu<T>;
    const int rc = ensure_lds((const void*)fn, lds);
    if (rc) return rc;
    ProfScope ps(st, PC_UNROLL);
    hipLaunchKernelGGL(fn, dim3(B), dim3(LQP_NT), lds, st, P, U);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

// ... and what is summed over the whole tape once every segment has been walked: Asbar / bsbar, Qsbar
template <typename T>
static int unroll_tape_finish_impl(hipStream_t st, int B, int n, int m, int iters, void* dQs, void* dAs, void* dbs, void* scratch,
                                   size_t scratch_bytes) {
    const int TT = iters + 1;
    UnrollLuCarve<T> c = carve_unroll_lu<T>(scratch, B, n, m, TT, true);
    if (scratch_bytes < c.bytes) return LQP_ERR_WORKSPACE;
    UnrollLuParams<T>& U = c.U;
    U.T_ = TT; U.dAs = (T*)dAs; U.dbs = (T*)dbs;
    if (m > 0) {
        ProfScope ps(st, PC_UNROLL);
        int slabs = (m * n + 255) / 256;
        if (slabs > 64) slabs = 64;
        hipLaunchKernelGGL(k_unroll_lu_eq<T>, dim3(B, slabs), dim3(256), 0, st, U, n, m);
    }
    if (dQs) {
        ProfScope ps(st, PC_UNROLL);
        const int tiles = (n + 63) / 64;
        hipLaunchKernelGGL(k_unroll_outer_any<T>, dim3(tiles, tiles, B), dim3(256), 0, st, (const T*)U.DX, (const T*)U.X, (T*)dQs, n, TT);
    }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

template <typename T>
static int unroll_backward_lu_impl(hipStream_t st, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes, int iters,
                                   const void* dl_dx, void* dQs, void* dps, void* dAs, void* dbs, void* dlbs, void* dubs, void* drho,
                                   void* dD, void* scratch, size_t scratch_bytes) {
    FwdLayout<T> L = carve_forward<T>((void*)fwd_workspace, B, n, m);
    if (fwd_workspace_bytes < L.bytes) return LQP_ERR_WORKSPACE;
    const FwdParams<T>& P = L.P;
    const int TT = iters + 1;
    UnrollLuCarve<T> c = carve_unroll_lu<T>(scratch, B, n, m, TT);
    if (scratch_bytes < c.bytes) return LQP_ERR_WORKSPACE;
    UnrollLuParams<T>& U = c.U;
    U.T_ = TT;
    U.g = (const T*)dl_dx;
    U.dps = (T*)dps; U.dlbs = (T*)dlbs; U.dubs = (T*)dubs; U.dD = (T*)dD; U.dAs = (T*)dAs; U.dbs = (T*)dbs; U.drho = (T*)drho;
    {
        const int lds = unroll_lu_lds_bytes<T>(P.Np);
        auto fn = k_unroll_sweep_lu<T>;
        const int rc = ensure_lds((const void*)fn, lds);
        if (rc) return rc;
        ProfScope ps(st, PC_UNROLL);
        hipLaunchKernelGGL(fn, dim3(B), dim3(LQP_NT), lds, st, P, U);
    }
    if (m > 0) {
        ProfScope ps(st, PC_UNROLL);
        int slabs = (m * n + 255) / 256;
        if (slabs > 64) slabs = 64;
        hipLaunchKernelGGL(k_unroll_lu_eq<T>, dim3(B, slabs), dim3(256), 0, st, U, n, m);
    }
    if (dQs) {
        ProfScope ps(st, PC_UNROLL);
        const int tiles = (n + 63) / 64;
        hipLaunchKernelGGL(k_unroll_outer_any<T>, dim3(tiles, tiles, B), dim3(256), 0, st, (const T*)U.DX, (const T*)U.X, (T*)dQs, n, TT);
    }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

}  // namespace

extern "C" {

int lqp_abi_version(void) { return LQP_ABI_VERSION; }

void lqp_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on = on != 0;
}

void lqp_profile_reset(void) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof_recs.clear();
    for (int i = 0; i < PC_COUNT; ++i) { g_prof_ms[i] = 0.0; g_prof_n[i] = 0; }
}

void lqp_debug_set_lu_counters(void* device_buf) { g_lu_dbg = (unsigned long long*)device_buf; }

// test aid: occupy `blocks` workgroups of 512 threads (one CU each when LDS-bound kernels run next to it) for `usec`
// microseconds on `stream` -- the co-residency tests run the two-workgroup schedules beside it
__global__ __launch_bounds__(512) void k_debug_spin(const unsigned long long ticks) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    smem[threadIdx.x] = 0;                  // (touch the LDS so that the allocation is real)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int lqp_debug_spin(void* stream, int blocks, int usec, int lds_bytes) {
    if (blocks < 1 || usec < 0 || usec > 2000000 || lds_bytes < 512) return LQP_ERR_INVALID;
    const int rc = ensure_lds((const void*)k_debug_spin, lds_bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(k_debug_spin, dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream, (unsigned long long)usec * 100ull);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

// test aid: the XCD every workgroup of a `blocks`-workgroup launch lands on (what the XCD-aware exchange asks at run time)
__global__ __launch_bounds__(512) void k_debug_xcd(int* __restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)my_xcd();
}
int lqp_debug_xcd(void* stream, int blocks, void* out_dev) {
    if (blocks < 1 || !out_dev) return LQP_ERR_INVALID;
    hipLaunchKernelGGL(k_debug_xcd, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (int*)out_dev);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

int lqp_debug_lu_inverse(void* stream, int dtype, int B, int N, const void* packed_buf, void* X_out) {
    if (bad_dims(dtype, B, N, 0) || !packed_buf || !X_out || N > (dtype == LQP_F32 ? 2048 : 1024)) return LQP_ERR_INVALID;
    const int K = round_up(N, LQP_NB) / LQP_NB;
    if (dtype == LQP_F32) {
        int* dest; float* packed;
        carve_packed<float>((void*)packed_buf, B, N, dest, packed);
        return launch_lu_inverse<float>((hipStream_t)stream, B, N, packed, packed_blocks(K) * LQP_BLK, dest, K * LQP_NB,
                                        (float*)X_out, (size_t)N * N, N, nullptr);
    }
    int* dest; double* packed;
    carve_packed<double>((void*)packed_buf, B, N, dest, packed);
    return launch_lu_inverse<double>((hipStream_t)stream, B, N, packed, packed_blocks(K) * LQP_BLK, dest, K * LQP_NB,
                                     (double*)X_out, (size_t)N * N, N, nullptr);
}

int lqp_profile_classes(void) { return PC_COUNT; }

const char* lqp_profile_class_name(int c) {
    static const char* names[PC_COUNT] = {"fwd_setup", "lu_factor", "pack", "admm_loop", "rho_update", "fwd_epilogue",
                                          "bwd_build", "packed_solve", "bwd_epilogue", "misc", "admm_loop_tail",
                                          "spd_inverse", "eq_correct", "bwd_cholesky", "unroll_backward", "unroll_scaling"};
    return (c >= 0 && c < PC_COUNT) ? names[c] : "?";
}

int lqp_profile_get(double* total_ms, long long* launches, int n) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (auto& r : g_prof_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return LQP_ERR_HIP;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return LQP_ERR_HIP;
        g_prof_ms[r.cls] += ms;
        g_prof_n[r.cls] += 1;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof_recs.clear();
    for (int i = 0; i < n && i < PC_COUNT; ++i) { total_ms[i] = g_prof_ms[i]; launches[i] = g_prof_n[i]; }
    return LQP_OK;
}

const char* lqp_status_string(int s) {
    switch (s) {
        case LQP_OK: return "ok";
        case LQP_ERR_INVALID: return "invalid argument";
        case LQP_ERR_WORKSPACE: return "workspace too small";
        case LQP_ERR_SINGULAR: return "singular matrix (exactly zero pivot)";
        case LQP_ERR_HIP: return "HIP runtime error";
        case LQP_ERR_TIMEOUT: return "in-kernel grid barrier timed out";
        case LQP_ERR_UNSUPPORTED: return "size not supported by this build (n + m <= 4096 in float32, 2048 in float64)";
        case LQP_ERR_NOT_SPD: return "matrix outside the symmetric x-update: repeat with linsolve = 1";
        default: return "unknown status";
    }
}

int lqp_boxqp_forward_layout(int dtype, int B, int n, int m, size_t* status_offset, size_t* status_bytes,
                             size_t* info_offset, size_t* info_bytes) {
    if (bad_dims(dtype, B, n, m) || !status_offset || !status_bytes || !info_offset || !info_bytes) return LQP_ERR_INVALID;
    char* base = (char*)4096;       // any non-null base: only offsets are used
    if (dtype == LQP_F32) {
        FwdLayout<float> L = carve_forward<float>(base, B, n, m);
        *status_offset = (char*)L.P.status - base; *info_offset = (char*)L.P.info - base;
    } else {
        FwdLayout<double> L = carve_forward<double>(base, B, n, m);
        *status_offset = (char*)L.P.status - base; *info_offset = (char*)L.P.info - base;
    }
    *status_bytes = sizeof(int) * ST_WORDS;
    *info_bytes = sizeof(int) * (size_t)B;
    return LQP_OK;
}

size_t lqp_boxqp_forward_workspace_bytes(int dtype, int B, int n, int m) {
    if (bad_dims(dtype, B, n, m)) return 0;
    return dtype == LQP_F32 ? carve_forward<float>(nullptr, B, n, m).bytes : carve_forward<double>(nullptr, B, n, m).bytes;
}

int lqp_boxqp_forward(void* stream, int dtype, int B, int n, int m, const void* Q, const void* p, const void* A,
                      const void* b, const void* lb, const void* ub, const lqp_boxqp_ctrl* ctrl, const void* rho_in,
                      void* x, void* z, void* u, void* lams, void* nus, void* rho_out, lqp_boxqp_stats* stats,
                      void* workspace, size_t workspace_bytes) {
    if (bad_dims(dtype, B, n, m) || !Q || !p || !lb || !ub || !ctrl || !x || !z || !u || !lams || !rho_out || !workspace)
        return LQP_ERR_INVALID;
    if (m > 0 && (!A || !b || !nus)) return LQP_ERR_INVALID;
    if (ctrl->rho_mode == 2 && !rho_in) return LQP_ERR_INVALID;
    if (ctrl->beta_mode == 2 && !ctrl->beta_in) return LQP_ERR_INVALID;
    if (ctrl->max_iters < 1) return LQP_ERR_INVALID;
    if (n + m > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int retry0 = (ctrl->reserved2 & 2) ? 4 : 0;      // (bit 1: the caller has seen a shared schedule time out -- nothing shared)
    if (dtype == LQP_F32)
        return forward_impl<float>(st, B, n, m, Q, p, A, b, lb, ub, ctrl, rho_in, x, z, u, lams, nus, rho_out, stats, workspace, workspace_bytes, retry0);
    return forward_impl<double>(st, B, n, m, Q, p, A, b, lb, ub, ctrl, rho_in, x, z, u, lams, nus, rho_out, stats, workspace, workspace_bytes, retry0);
}

int lqp_boxqp_forward_finish(void* stream, int B, int max_iters, int check_solved, const void* host_report,
                             lqp_boxqp_stats* stats) {
    if (!host_report || !stats || B < 1) return LQP_ERR_INVALID;
    if (stats->mode_used != 4) return LQP_OK;               // (the forward call waited itself)
    const int rc = collect_report((hipStream_t)stream, (const int*)host_report, true, B, max_iters,
                                  check_solved < 1 ? 1 : check_solved, stats);
    return rc == LQP_RETRY_LU ? LQP_ERR_NOT_SPD : rc;
}

// ---- unroll=True: backward through the unrolled loop (lqp_unroll.hpp) ----
struct UnrollCarve { UnrollParams U; size_t bytes; };
static UnrollCarve carve_unroll(void* ws, int B, int n, int m, int T) {
    UnrollCarve c;
    memset(&c.U, 0, sizeof(c.U));
    Carver cv(ws);
    c.U.X = cv.take<float>((size_t)B * T * n);
    c.U.W = cv.take<float>((size_t)B * T * n);
    c.U.DX = cv.take<float>((size_t)B * T * n);
    c.U.NU = cv.take<float>((size_t)B * T * (m > 0 ? m : 1));
    c.U.MK = cv.take<signed char>((size_t)B * T * n);
    c.bytes = cv.off + kAlign;
    return c;
}

size_t lqp_boxqp_unroll_backward_workspace_bytes(int B, int n, int m, int iters) {
    if (B < 1 || n < 1 || m < 0 || iters < 0) return 0;
    return carve_unroll(nullptr, B, n, m, iters + 1).bytes;
}

int lqp_boxqp_unroll_backward(void* stream, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes,
                              int iters, const void* dl_dx, void* dQs, void* dps, void* dAs, void* dbs, void* dlbs,
                              void* dubs, void* drho, void* dD, void* scratch, size_t scratch_bytes) {
    if (B < 1 || n < 1 || m < 0 || iters < 0 || !fwd_workspace || !dl_dx || !dps || !dlbs || !dubs || !drho || !dD || !scratch)
        return LQP_ERR_INVALID;
    if (m > 0 && (!dAs || !dbs)) return LQP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    FwdLayout<float> L = carve_forward<float>((void*)fwd_workspace, B, n, m);
    if (fwd_workspace_bytes < L.bytes) return LQP_ERR_WORKSPACE;
    const FwdParams<float>& P = L.P;
    if (P.Ks > SPD_BIGK || m > SPD_MAXM) return LQP_ERR_UNSUPPORTED;      // (what the symmetric x-update takes)
    const int T = iters + 1;
    UnrollCarve c = carve_unroll(scratch, B, n, m, T);
    if (scratch_bytes < c.bytes) return LQP_ERR_WORKSPACE;
    UnrollParams& U = c.U;
    U.T = T;
    U.rl = unroll_lds_blocks(m, P.Ks);
    U.g = (const float*)dl_dx;
    U.dps = (float*)dps; U.dlbs = (float*)dlbs; U.dubs = (float*)dubs; U.dD = (float*)dD;
    U.dAs = (float*)dAs; U.dbs = (float*)dbs; U.drho = (float*)drho;
    // two workgroups per QP when half the chip would idle (the split loop's products, lqp_unroll.hpp: k_unroll_sweep_split)
    bool split_done = false;
    if (P.xchg && P.Ks >= 5 && P.Ks <= SPD_MAXK && knobs().unroll_split != 0) {
        void (*fn2)(const FwdParams<float>, const UnrollParams, const unsigned int) =
            P.Ks == 8 ? (m <= 1 ? k_unroll_sweep_split<8, 1> : k_unroll_sweep_split<8, SPD_MAXM>)
            : P.Ks == 7 ? k_unroll_sweep_split<7, SPD_MAXM> : P.Ks == 6 ? k_unroll_sweep_split<6, SPD_MAXM> : k_unroll_sweep_split<5, SPD_MAXM>;
        const int lds2 = P.Ks == 8 ? unroll_split_lds_bytes<8>(m) : P.Ks == 7 ? unroll_split_lds_bytes<7>(m)
                       : P.Ks == 6 ? unroll_split_lds_bytes<6>(m) : unroll_split_lds_bytes<5>(m);
        int dev = 0, cus = 0, per_cu = 0;
        if (current_device_cus(&dev, &cus) && ensure_lds((const void*)fn2, lds2) == LQP_OK &&
            blocks_per_cu(&per_cu, fn2, 512, lds2, dev) && per_cu >= 1 && shared_grid(B, 2) <= cus * per_cu) {
            static std::atomic<unsigned int> run{1u};
            ProfScope ps(st, PC_UNROLL);
            hipLaunchKernelGGL(fn2, dim3((knobs().dbg_loop_absent & 4) ? B : shared_grid(B, 2)), dim3(512), lds2, st, P, U, run.fetch_add(1u));
            split_done = true;
        }
    }
    if (!split_done) {
        const int lds = unroll_lds_bytes(m, P.Ks, U.rl);
        auto sweep_fn = m <= 1 ? k_unroll_sweep<1> : k_unroll_sweep<SPD_MAXM>;
        int rc = ensure_lds((const void*)sweep_fn, lds);
        if (rc) return rc;
        ProfScope ps(st, PC_UNROLL);
        hipLaunchKernelGGL(sweep_fn, dim3(B), dim3(LQP_NT), lds, st, P, U);
    }
    if (dQs) {
        ProfScope ps(st, PC_UNROLL);
        const int tiles = (n + 63) / 64;
        hipLaunchKernelGGL(k_unroll_outer<>, dim3(tiles, tiles, B), dim3(256), 0, st, (const float*)U.DX, (const float*)U.X,
                           (float*)dQs, n, T);
    }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

size_t lqp_boxqp_unroll_backward_lu_workspace_bytes(int dtype, int B, int n, int m, int iters) {
    if (bad_dims(dtype, B, n, m) || iters < 0) return 0;
    return dtype == LQP_F32 ? carve_unroll_lu<float>(nullptr, B, n, m, iters + 1).bytes : carve_unroll_lu<double>(nullptr, B, n, m, iters + 1).bytes;
}

int lqp_boxqp_unroll_backward_lu(void* stream, int dtype, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes,
                                 int iters, const void* dl_dx, void* dQs, void* dps, void* dAs, void* dbs, void* dlbs, void* dubs,
                                 void* drho, void* dD, void* scratch, size_t scratch_bytes) {
    if (bad_dims(dtype, B, n, m) || iters < 0 || !fwd_workspace || !dl_dx || !dps || !dlbs || !dubs || !drho || !dD || !scratch)
        return LQP_ERR_INVALID;
    if (m > 0 && (!dAs || !dbs)) return LQP_ERR_INVALID;
    if (n + m > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LQP_F32)
        return unroll_backward_lu_impl<float>(st, B, n, m, fwd_workspace, fwd_workspace_bytes, iters, dl_dx, dQs, dps, dAs, dbs, dlbs, dubs,
                                              drho, dD, scratch, scratch_bytes);
    return unroll_backward_lu_impl<double>(st, B, n, m, fwd_workspace, fwd_workspace_bytes, iters, dl_dx, dQs, dps, dAs, dbs, dlbs, dubs,
                                           drho, dD, scratch, scratch_bytes);
}

size_t lqp_boxqp_unroll_tape_workspace_bytes(int dtype, int B, int n, int m, int iters) {
    if (bad_dims(dtype, B, n, m) || iters < 0) return 0;
    return dtype == LQP_F32 ? carve_unroll_lu<float>(nullptr, B, n, m, iters + 1, true).bytes
                            : carve_unroll_lu<double>(nullptr, B, n, m, iters + 1, true).bytes;
}

int lqp_boxqp_unroll_tape_segment(void* stream, int dtype, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes,
                                  int iters, int k0, int k1, int mode, const void* packed_buf, const void* rho, void* state, int inj_k,
                                  const void* inj, const void* dl_dx, void* dps, void* dlbs, void* dubs, void* drho, void* dD,
                                  void* scratch, size_t scratch_bytes, void** z_rows, void** u_rows, void** x_rows) {
    if (bad_dims(dtype, B, n, m) || iters < 0 || !fwd_workspace || !scratch || k0 < 0 || k1 > iters + 1 || (mode & ~3) != 0)
        return LQP_ERR_INVALID;
    if (k1 > k0 && (mode & 2) && (!dl_dx || !dps || !dlbs || !dubs || !drho || !dD)) return LQP_ERR_INVALID;
    if (n + m > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LQP_F32)
        return unroll_tape_segment_impl<float>(st, B, n, m, fwd_workspace, fwd_workspace_bytes, iters, k0, k1, mode, packed_buf, rho, state,
                                               inj_k, inj, dl_dx, dps, dlbs, dubs, drho, dD, scratch, scratch_bytes, z_rows, u_rows, x_rows);
    return unroll_tape_segment_impl<double>(st, B, n, m, fwd_workspace, fwd_workspace_bytes, iters, k0, k1, mode, packed_buf, rho, state,
                                            inj_k, inj, dl_dx, dps, dlbs, dubs, drho, dD, scratch, scratch_bytes, z_rows, u_rows, x_rows);
}

int lqp_boxqp_unroll_tape_finish(void* stream, int dtype, int B, int n, int m, int iters, void* dQs, void* dAs, void* dbs, void* scratch,
                                 size_t scratch_bytes) {
    if (bad_dims(dtype, B, n, m) || iters < 0 || !scratch || (m > 0 && (!dAs || !dbs))) return LQP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    return dtype == LQP_F32 ? unroll_tape_finish_impl<float>(st, B, n, m, iters, dQs, dAs, dbs, scratch, scratch_bytes)
                            : unroll_tape_finish_impl<double>(st, B, n, m, iters, dQs, dAs, dbs, scratch, scratch_bytes);
}

int lqp_unroll_scale_colmax(void* stream, int B, int n, const void* Q, void* colmax, void* argmax, void* count) {
    if (B < 0 || n < 1 || !Q || !colmax || !argmax || !count) return LQP_ERR_INVALID;
    if (B == 0) return LQP_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, PC_UNROLL_SCALE);
    int dev = 0, cus = 0;
    const int slabs = (current_device_cus(&dev, &cus) && 2 * B <= cus && n >= 128) ? 2 : 1;      // column halves when half the chip is idle
    hipLaunchKernelGGL(k_unroll_scale_colmax<>, dim3(B, slabs), dim3(LQP_NT), 0, st, (const float*)Q, n, (float*)colmax, (int*)argmax, (int*)count);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

int lqp_unroll_scale_grad_slabs(int B, int n) {
    (void)n;
    int dev = 0, cus = 0;
    if (B < 1 || !current_device_cus(&dev, &cus)) return 1;
    int s = (8 * cus) / B;                       // 256-thread workgroups: eight per CU fill it
    return s < 1 ? 1 : (s > 32 ? 32 : s);
}

int lqp_unroll_scale_grad(void* stream, int B, int n, const void* Q, const void* d, const void* s, void* G, void* parts, int slabs) {
    if (B < 0 || n < 1 || !Q || !G || !parts || slabs < 1 || slabs > 64) return LQP_ERR_INVALID;
    if (B == 0) return LQP_OK;
    hipStream_t st = (hipStream_t)stream;
    const int lds = 4 * n * (int)sizeof(float);
    if (lds > 64 * 1024) return LQP_ERR_UNSUPPORTED;
    int rc = ensure_lds((const void*)k_unroll_scale_grad<>, lds);
    if (rc) return rc;
    ProfScope ps(st, PC_UNROLL_SCALE);
    hipLaunchKernelGGL(k_unroll_scale_grad<>, dim3(B, slabs), dim3(256), lds, st, (const float*)Q, (const float*)d, (const float*)s,
                       (float*)G, n, (float*)parts);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

int lqp_unroll_scale_vectors(void* stream, int B, int n, int m, int phase, int has_box, int beta_given, double beta_value,
                             const void* colmax, const void* p, const void* A, const void* b, const void* lb, const void* ub,
                             const void* g_ps, const void* g_As, const void* g_bs, const void* g_lbs, const void* g_ubs, const void* g_D,
                             const void* parts, int nparts, void* d_out, void* dp, void* dA, void* db, void* dlb, void* dub,
                             void* g_colmax) {
    if (B < 0 || n < 1 || m < 0 || !colmax || (phase != 0 && phase != 1)) return LQP_ERR_INVALID;
    if (phase == 0 && !d_out) return LQP_ERR_INVALID;
    if (phase == 1 && (!p || !g_colmax || (m > 0 && (!A || !b || !g_As)) || (has_box && (!lb || !ub)) || (nparts > 0 && !parts)))
        return LQP_ERR_INVALID;
    if (B == 0) return LQP_OK;
    const int lds = unroll_scale_vectors_lds_bytes(n, m);
    if (lds > 160 * 1024) return LQP_ERR_UNSUPPORTED;
    int rc = ensure_lds((const void*)k_unroll_scale_vectors<>, lds);
    if (rc) return rc;
    ScaleVecParams P;
    memset(&P, 0, sizeof(P));
    P.n = n; P.m = m; P.phase = phase; P.has_box = has_box; P.beta_given = beta_given; P.nparts = nparts; P.beta_value = (float)beta_value;
    P.cn = (const float*)colmax; P.p = (const float*)p; P.A = (const float*)A; P.b = (const float*)b; P.lb = (const float*)lb; P.ub = (const float*)ub;
    P.gps = (const float*)g_ps; P.gAs = (const float*)g_As; P.gbs = (const float*)g_bs; P.glbs = (const float*)g_lbs; P.gubs = (const float*)g_ubs;
    P.gD = (const float*)g_D; P.parts = (const float*)parts;
    P.d_out = (float*)d_out; P.dp = (float*)dp; P.dA = (float*)dA; P.db = (float*)db; P.dlb = (float*)dlb; P.dub = (float*)dub; P.gcn = (float*)g_colmax;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, PC_UNROLL_SCALE);
    hipLaunchKernelGGL(k_unroll_scale_vectors<>, dim3(B), dim3(LQP_NT), lds, st, P);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

int lqp_unroll_scale_scatter(void* stream, int B, int n, const void* Q, const void* colmax, const void* argmax, const void* count,
                             const void* g_colmax, void* G) {
    if (B < 0 || n < 1 || !Q || !colmax || !argmax || !count || !g_colmax || !G) return LQP_ERR_INVALID;
    if (B == 0) return LQP_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, PC_UNROLL_SCALE);
    hipLaunchKernelGGL(k_unroll_scale_scatter<>, dim3(B), dim3(LQP_NT), 0, st, (const float*)Q, (const float*)colmax, (const int*)argmax,
                       (const int*)count, (const float*)g_colmax, (float*)G, n);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

int lqp_boxqp_last_residuals(void* stream, int dtype, int B, int n, int m, const void* workspace, size_t workspace_bytes,
                             void* primal_out, void* dual_out) {
    if (bad_dims(dtype, B, n, m) || !workspace) return LQP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return LQP_OK;
    if (dtype == LQP_F32) {
        FwdLayout<float> L = carve_forward<float>(const_cast<void*>(workspace), B, n, m);
        if (workspace_bytes < L.bytes) return LQP_ERR_WORKSPACE;
        hipLaunchKernelGGL(k_copy_residuals<float>, dim3((B + 255) / 256), dim3(256), 0, st, L.P.scal, (float*)primal_out,
                           (float*)dual_out, B);
    } else {
        FwdLayout<double> L = carve_forward<double>(const_cast<void*>(workspace), B, n, m);
        if (workspace_bytes < L.bytes) return LQP_ERR_WORKSPACE;
        hipLaunchKernelGGL(k_copy_residuals<double>, dim3((B + 255) / 256), dim3(256), 0, st, L.P.scal, (double*)primal_out,
                           (double*)dual_out, B);
    }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

int lqp_boxqp_check_trace(void* stream, int dtype, int B, int n, int m, const void* workspace, size_t workspace_bytes,
                          int n_checks, void* trace_out) {
    if (bad_dims(dtype, B, n, m) || !workspace || !trace_out || n_checks < 0) return LQP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || n_checks == 0) return LQP_OK;
    const unsigned int* src;
    if (dtype == LQP_F32) {
        FwdLayout<float> L = carve_forward<float>(const_cast<void*>(workspace), B, n, m);
        if (workspace_bytes < L.bytes) return LQP_ERR_WORKSPACE;
        src = L.vtrace_area;
    } else {
        FwdLayout<double> L = carve_forward<double>(const_cast<void*>(workspace), B, n, m);
        if (workspace_bytes < L.bytes) return LQP_ERR_WORKSPACE;
        src = L.vtrace_area;
    }
    const int words = 2 * (n_checks < kRing ? n_checks : kRing);
    hipLaunchKernelGGL(k_copy_trace<>, dim3((words + 255) / 256), dim3(256), 0, st, src, (float*)trace_out, words);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

size_t lqp_boxqp_backward_fp_workspace_bytes(int dtype, int B, int n, int m) {
    if (bad_dims(dtype, B, n, m)) return 0;
    if (dtype == LQP_F32) { BwdParams<float> P; return carve_backward<float>(nullptr, B, n, m, P); }
    BwdParams<double> P; return carve_backward<double>(nullptr, B, n, m, P);
}

int lqp_boxqp_backward_fp_prefactor(void* stream, int dtype, int B, int n, int m, const void* x, const void* u, const void* Q,
                                    const void* A, const void* lb, const void* ub, void* workspace, size_t workspace_bytes,
                                    int linsolve, void* host_report) {
    if (bad_dims(dtype, B, n, m) || !x || !u || !Q || !lb || !ub || !workspace) return LQP_ERR_INVALID;
    if (m > 0 && !A) return LQP_ERR_INVALID;
    if (n + m > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    if (dtype == LQP_F32)
        return backward_impl<float>((hipStream_t)stream, B, n, m, nullptr, x, u, nullptr, nullptr, Q, A, lb, ub, 1, 1.0, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                                    linsolve == 2 ? 2 : 1, host_report, 0, 1);
    return backward_impl<double>((hipStream_t)stream, B, n, m, nullptr, x, u, nullptr, nullptr, Q, A, lb, ub, 1, 1.0, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, 1,
                                 host_report, 0, 1);
}

int lqp_boxqp_backward_fp(void* stream, int dtype, int B, int n, int m, const void* dl_dz, const void* x, const void* u,
                          const void* lams, const void* nus, const void* Q, const void* A, const void* lb, const void* ub,
                          int rho_mode, double rho_value, const void* rho_in, void* dQ, void* dp, void* dA, void* db,
                          void* dlb, void* dub, int32_t* fail_index, void* workspace, size_t workspace_bytes, int linsolve,
                          void* host_report) {
    if (bad_dims(dtype, B, n, m) || !dl_dz || !x || !u || !lams || !Q || !lb || !ub || !workspace) return LQP_ERR_INVALID;
    if (m > 0 && (!A || !nus)) return LQP_ERR_INVALID;
    if (rho_mode != 1 && rho_mode != 2) return LQP_ERR_INVALID;
    if (rho_mode == 2 && !rho_in) return LQP_ERR_INVALID;
    if (n + m > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LQP_F32)
        return backward_impl<float>(st, B, n, m, dl_dz, x, u, lams, nus, Q, A, lb, ub, rho_mode, rho_value, rho_in, dQ, dp, dA, db, dlb, dub, fail_index, workspace, workspace_bytes, linsolve, host_report);
    return backward_impl<double>(st, B, n, m, dl_dz, x, u, lams, nus, Q, A, lb, ub, rho_mode, rho_value, rho_in, dQ, dp, dA, db, dlb, dub, fail_index, workspace, workspace_bytes,
                                 1 | (linsolve & (LQP_BWD_PREFACTORED | LQP_BWD_REPORTED)), host_report);
}

int lqp_boxqp_backward_kkt(void* stream, int dtype, int B, int n, int m, const void* dl_dz, const void* x,
                           const void* lams, const void* nus, const void* Q, const void* A, const void* lb, const void* ub,
                           void* dQ, void* dp, void* dA, void* db, void* dlb, void* dub, int32_t* fail_index, void* workspace,
                           size_t workspace_bytes, int linsolve, void* host_report) {
    if (bad_dims(dtype, B, n, m) || !dl_dz || !x || !lams || !Q || !lb || !ub || !workspace) return LQP_ERR_INVALID;
    if (m > 0 && (!A || !nus)) return LQP_ERR_INVALID;
    if (n + m > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LQP_F32)
        return backward_impl<float>(st, B, n, m, dl_dz, x, nullptr, lams, nus, Q, A, lb, ub, 1, 1.0, nullptr, dQ, dp, dA, db, dlb,
                                    dub, fail_index, workspace, workspace_bytes, linsolve, host_report, 1);
    return backward_impl<double>(st, B, n, m, dl_dz, x, nullptr, lams, nus, Q, A, lb, ub, 1, 1.0, nullptr, dQ, dp, dA, db, dlb,
                                 dub, fail_index, workspace, workspace_bytes, 1, host_report, 1);
}

size_t lqp_spd_inverse_workspace_bytes(int dtype, int B, int n) {
    if (dtype != LQP_F32 || B < 0 || n < 1) return 0;
    const int Ks = round_up(n, LQP_NB) / LQP_NB;
    // (above 512: + the panel scratch of wg_spd_sweep_big)
    return (size_t)B * (sym_blocks(Ks) + (Ks > SPD_MAXK ? Ks - 1 : 0)) * LQP_BLK * sizeof(float) + 2 * kAlign;
}

int lqp_spd_inverse_batched(void* stream, int dtype, int B, int n, const void* K_in, void* Kinv_out, int32_t* info_out,
                            void* workspace, size_t workspace_bytes) {
    if (dtype != LQP_F32) return LQP_ERR_UNSUPPORTED;
    if (B < 0 || n < 1 || !K_in || !Kinv_out || !info_out || !workspace) return LQP_ERR_INVALID;
    const int Ks = round_up(n, LQP_NB) / LQP_NB;
    if (Ks > SPD_BIGK) return LQP_ERR_UNSUPPORTED;
    if (workspace_bytes < lqp_spd_inverse_workspace_bytes(dtype, B, n)) return LQP_ERR_WORKSPACE;
    if (B == 0) return LQP_OK;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace);
    float* Hs = c.take<float>((size_t)B * sym_blocks(Ks) * LQP_BLK);
    float* Yg = Ks > SPD_MAXK ? c.take<float>((size_t)B * (Ks - 1) * LQP_BLK) : nullptr;
    const int lds = spd_lds_bytes(Ks > SPD_MAXK ? SPD_MAXK : Ks);
    auto dense_fn = Ks > SPD_MAXK ? k_spd_inverse_dense<2> : k_spd_inverse_dense<1>;
    int rc = ensure_lds((const void*)dense_fn, lds);
    if (rc) return rc;
    { ProfScope ps(st, PC_SPD_INV);
      hipLaunchKernelGGL(dense_fn, dim3(B), dim3(LQP_NT), lds, st, (const float*)K_in, (float*)Kinv_out, Hs,
                         (int*)info_out, n, Ks, Yg); }
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

size_t lqp_lu_factor_workspace_bytes(int dtype, int B, int N) {
    if (bad_dims(dtype, B, N, 0)) return 0;
    int* piv; unsigned long long* scr;
    if (dtype == LQP_F32) { float* M; return carve_lu<float>(nullptr, B, N, M, piv, scr); }
    double* M; return carve_lu<double>(nullptr, B, N, M, piv, scr);
}

int lqp_lu_factor_batched(void* stream, int dtype, int B, int N, void* M_inout, int32_t* piv_out, int32_t* info_out,
                          void* workspace, size_t workspace_bytes) {
    if (bad_dims(dtype, B, N, 0) || !M_inout || !piv_out || !info_out || !workspace) return LQP_ERR_INVALID;
    if (N > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    return dtype == LQP_F32 ? lu_factor_impl<float>(st, B, N, M_inout, piv_out, info_out, workspace, workspace_bytes)
                            : lu_factor_impl<double>(st, B, N, M_inout, piv_out, info_out, workspace, workspace_bytes);
}

size_t lqp_lu_packed_bytes(int dtype, int B, int N) {
    if (bad_dims(dtype, B, N, 0)) return 0;
    return dtype == LQP_F32 ? packed_bytes_t<float>(B, N) : packed_bytes_t<double>(B, N);
}
size_t lqp_lu_solve_workspace_bytes(int dtype, int B, int N) { return lqp_lu_packed_bytes(dtype, B, N); }

int lqp_lu_pack(void* stream, int dtype, int B, int N, const void* LU, const int32_t* piv, void* packed) {
    if (bad_dims(dtype, B, N, 0) || !LU || !piv || !packed) return LQP_ERR_INVALID;
    if (N > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    return dtype == LQP_F32 ? lu_pack_impl<float>(st, B, N, LU, piv, packed) : lu_pack_impl<double>(st, B, N, LU, piv, packed);
}

int lqp_lu_solve_packed(void* stream, int dtype, int B, int N, int k, const void* packed, void* rhs_inout) {
    if (bad_dims(dtype, B, N, 0) || k < 1 || !packed || !rhs_inout) return LQP_ERR_INVALID;
    if (N > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    return dtype == LQP_F32 ? lu_solve_packed_impl<float>(st, B, N, k, packed, rhs_inout)
                            : lu_solve_packed_impl<double>(st, B, N, k, packed, rhs_inout);
}

int lqp_lu_solve_batched(void* stream, int dtype, int B, int N, int k, const void* LU, const int32_t* piv, void* rhs_inout,
                         void* workspace, size_t workspace_bytes) {
    if (!workspace || workspace_bytes < lqp_lu_packed_bytes(dtype, B, N)) return workspace ? LQP_ERR_WORKSPACE : LQP_ERR_INVALID;
    int rc = lqp_lu_pack(stream, dtype, B, N, LU, piv, workspace);
    if (rc) return rc;
    return lqp_lu_solve_packed(stream, dtype, B, N, k, workspace, rhs_inout);
}

size_t lqp_kkt_solve_workspace_bytes(int dtype, int B, int n, int m) {
    if (bad_dims(dtype, B, n, m)) return 0;
    int *piv, *dest, *info;
    if (dtype == LQP_F32) { float *M, *pk, *rhs; return carve_kkt<float>(nullptr, B, n, m, M, pk, rhs, piv, dest, info); }
    double *M, *pk, *rhs; return carve_kkt<double>(nullptr, B, n, m, M, pk, rhs, piv, dest, info);
}

int lqp_kkt_solve(void* stream, int dtype, int B, int n, int m, const void* Q, const void* p, const void* A, const void* b,
                  void* x, void* nus, int32_t* fail_index, void* workspace, size_t workspace_bytes) {
    if (bad_dims(dtype, B, n, m) || !Q || !p || !x || !workspace) return LQP_ERR_INVALID;
    if (m > 0 && (!A || !b || !nus)) return LQP_ERR_INVALID;
    if (n + m > max_rows(dtype)) return LQP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    return dtype == LQP_F32 ? kkt_solve_impl<float>(st, B, n, m, Q, p, A, b, x, nus, fail_index, workspace, workspace_bytes)
                            : kkt_solve_impl<double>(st, B, n, m, Q, p, A, b, x, nus, fail_index, workspace, workspace_bytes);
}

int lqp_qp_outer_grads(void* stream, int dtype, int B, int n, int m, const void* dx, const void* x, const void* dnu,
                       const void* nus, void* dQ, void* dA) {
    if (bad_dims(dtype, B, n, m) || !dx || !x) return LQP_ERR_INVALID;
    if (m > 0 && dA && (!dnu || !nus)) return LQP_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LQP_F32)
        hipLaunchKernelGGL(k_outer_grads<float>, dim3(B), dim3(LQP_NT), 0, st, (const float*)dx, (const float*)x,
                           (const float*)dnu, (const float*)nus, n, m, (float*)dQ, (float*)dA);
    else
        hipLaunchKernelGGL(k_outer_grads<double>, dim3(B), dim3(LQP_NT), 0, st, (const double*)dx, (const double*)x,
                           (const double*)dnu, (const double*)nus, n, m, (double*)dQ, (double*)dA);
    return hipGetLastError() == hipSuccess ? LQP_OK : LQP_ERR_HIP;
}

}  // extern "C"
