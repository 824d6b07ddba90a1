// libkbdm_hip.so - host side of the C ABI declared in include/kbdm_hip.h.
// Plans the batch (sort by cost, carve the workspace, chunk if it would not fit), enqueues
// the kernels of kbdm_kernels.hpp on one HIP stream and moves results.  No CPU compute path
// exists here: without a gfx950 device every entry point fails with KBDM_E_NODEVICE/HIP.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <limits>
#include <thread>
#include <atomic>
#include <chrono>
#include <vector>

#include "../../include/kbdm_hip.h"
#include "kbdm_kernels.hpp"
#include "kbdm_next.hpp"
#include "kbdm_cluster.hpp"

using namespace kb;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            return fail(KBDM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

// inside a `do { ... } while (0)` block with an `int r`: record the failure and leave the block (cleanup follows it)
#define HIPTRY(expr)                                                                        \
    {                                                                                       \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            r = fail(KBDM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));       \
            break;                                                                          \
        }                                                                                   \
    }

// device allocations of a stage entry point, released on every path out of it
struct DevBufs {
    std::vector<void*> ptrs;
    ~DevBufs() { for (void* p : ptrs) if (p) hipFree(p); }
    template <class T> hipError_t alloc(T** out, size_t bytes) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) { ptrs.push_back(p); *out = static_cast<T*>(p); }
        return e;
    }
};

// Test hooks (kbdm_debug_force_status): status bits that kbdm_plan_download / kbdm_plan_collect OR into every member's word
// - always / in the next collected run only.  Set explicitly through the C ABI by the tests of the status contract; no
// environment variable is read on the result path.
std::atomic<int> g_force_status{0}, g_force_status_once{0};

int env_int(const char* name, int def) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : def;
}

constexpr int LDS_MAX = 160 * 1024;

const char* kStageNames[KBDM_NSTAGES] = {"k_hankel",  "k_svd_fac", "k_gen(Q,P)", "k_dc_tree", "k_dc_final", "k_dc_sv",
                                         "k_gemm<1>", "k_gemm<2>", "k_hess",     "k_gen(Qh)",   "k_hqr",         "k_invit",
                                         "k_gemm<3>", "k_gemm<4>", "k_gemm<5>",  "k_epilogue"};

}  // namespace

// A lane = one in-order pipeline (main stream + side stream).  The members of a batch are split
// over the lanes by size, so that the throughput-bound stages of the small members run underneath
// the latency-bound stages (one CU per member) of the large ones.
struct Lane {
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // side stream: work that is independent of the main chain
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_done = nullptr;
};

#define KB_MAX_LANES 8
#define KB_QUEUE_WORDS 64   // work-queue counters behind the per-member words of d_iwork (one per chunk)

struct kbdm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;             // = lanes[0].stream: the stream the C ABI synchronises on
    Lane lanes[KB_MAX_LANES];
    hipEvent_t ev_start = nullptr;
    int nlanes = 2;
    double lane0_frac = 0.5;   // share of the batch's cost (sum of m^3) that lane 0 takes when there are two lanes
    int nt_fac = 1024;    // threads per workgroup: bidiagonalisation / Hessenberg kernels
    int nt_invit = 1024;
    int split_gen = 32;   // workgroups per item and matrix in k_gen (columns of one matrix are independent)
    int split_invit = 8;  // workgroups per item in k_invit
    int last_eig_fallbacks = 0;   // members the Aberth path handed to the QR iteration in the last kbdm_eig_batch call
    int invit_reg = 1;    // inverse iteration with register-resident vectors for l <= 512 (0: the LDS form)
    int invit_wpb = 4;    // wavefronts per workgroup of k_invit_reg<2>, <4> (one per SIMD: the 256 VGPRs of a wavefront leave room for a
                          // workgroup of another kernel - an MFMA-bound k_ab_iter tile beside this VALU-bound solve)
    int invit_wpb8 = 4;   // ... of k_invit_reg<8>
    int team_hqr = 1;     // large members of lane 0: chase workgroup + helper workgroup (k_hqr_team)
    int team_min_l = 192; // smallest l that gets a team
    int hqr_wgs = -1;     // workgroups of the solo k_hqr launch (members are taken from a queue, largest first):
                          // -1 = as many as the launch's work needs to last no longer than its largest member,
                          // 0 = one workgroup per member, N = fixed
    int gen_wy = 1;       // explicit Q / P / Qh of large members by blocked compact-WY accumulation on MFMA (0: k_gen for all)
    int blocked = 1;      // blocked (panel + MFMA update) reductions; 0: unblocked kernels only (debugging)
    int hqr_prof = 0;     // KBDM_HQR_PROF: cycle-counter dump of the QR iteration (diagnostic, synchronous)
    int nb_hqr2 = 8;      // bulges in flight (two shifts each) of the second-generation iteration
    int win_hqr2 = KB2_WIN_DEV;   // its LDS window (fixed: the device chase is compiled for it)
    int eig_ab = 1;       // eigenvalues by divide-and-conquer Ehrlich-Aberth (kb_aberth.hpp), QR iteration as the fallback;
                          // 0: QR iteration for every member (KBDM_EIG_AB)
    int team_max = 112;   // teams in flight over all lanes: 2 workgroups each, one workgroup per CU, all resident
    int panel_T = 1;      // cooperative panels (kb_team.hpp): workgroups per member of lane 0 in the two panel kernels ...
    int panel_budget = 64;    // ... as long as all teams of the launch fit this many workgroups (they wait for each other, so
                          // they must all be resident: the sum over the contexts of a process must stay below the CU count)
    int panel_T_all = 0;  // KBDM_PANEL_T_ALL=1: teams on every lane (the budget then counts per lane)
    int ab_tail = 1;      // the root level's tail on the wavefront-per-root kernel (KBDM_AB_TAIL=0: tile kernel throughout)
    int ab_dbg = 0;       // KBDM_AB_DBG, read once when the context is created: 8 = phase timers of k_ab_iter (tools/ab_phases.py);
                          // the bits that skip work (2, 4: timing experiments, wrong results) exist in -DKBDM_DEBUG_BUILD libraries only
    double ws_budget_gib = 96.0;
    // multi-GPU: RCCL communicator (one per context) and the device buffers of the packed gather
    void* comm = nullptr;
    bool comm_owned = false;      // false: borrowed from another context of this process (kbdm_comm_attach)
    int comm_world = 0, comm_rank = 0;
    hipStream_t comm_stream = nullptr;        // the gather runs here, behind an event of the plan's stream
    hipEvent_t ev_packed = nullptr, ev_gathered = nullptr;
    bool gather_pending = false;
    int* d_gather_bad = nullptr;              // set by k_check_trailers when a received block is not the one this step expects
    char* d_pack = nullptr;
    char* d_gather = nullptr;
    size_t pack_cap = 0, gather_cap = 0;
};

struct Chunk {
    int first = 0, count = 0;   // range in sorted order
    int mmax = 0, lmax = 0;
    int group = 0;              // workspace generation: groups run one after the other (they share the arena)
    int lane = 0;               // chunks of one group run concurrently, one per lane
    std::vector<hipEvent_t> ev;
    // opt-in per-kernel timers (KBDM_MODE_KERNEL_TIMERS): pairs of events around the launches of a kernel class
    std::vector<hipEvent_t> kev[KBDM_NKCLASSES];
    int kused[KBDM_NKCLASSES] = {0};       // events used by the current run (2 per bracket)
    int klaunches[KBDM_NKCLASSES] = {0};   // kernel launches inside the brackets
};

const char* kKernelClassNames[KBDM_NKCLASSES] = {"k_hankel", "k_bidiag_panel_team", "k_trail_update", "k_hess_panel_team",
                                                 "k_hess_z", "k_hess_update", "k_ab_iter", "k_wy_apply"};

struct kbdm_plan {
    kbdm_ctx* ctx = nullptr;
    int S = 0, N = 0, B = 0, p = 1;
    double q = 0.0, dwell = 0.0;
    std::vector<KbItem> items;     // caller order
    std::vector<int> perm;         // sorted position -> item index (descending cost)
    std::vector<Chunk> chunks;
    std::vector<int64_t> line_off, sv_off;
    int64_t total_lines = 0, total_sv = 0;
    size_t arena_elems = 0, varena_elems = 0, dc_elems = 0;
    double* d_dc = nullptr;        // divide-and-conquer workspace (doubles): SVD tree, then the Aberth panels
    int* d_abstat = nullptr;       // working tiles per (step, iteration) of the Aberth path: 16 x KB_AB_BUDGET counters
    int* d_needqr = nullptr;       // per member: 1 = the Ehrlich-Aberth path declined, the QR iteration solves it
    int* d_iwork = nullptr;
    cd* d_signals = nullptr;
    KbItem* d_items = nullptr;
    int* d_perm = nullptr;
    cd* d_arena = nullptr;
    double* d_varena = nullptr;
    double* d_lines = nullptr;
    double* d_sv = nullptr;
    cd* d_mu = nullptr;
    unsigned char* d_keep = nullptr;
    int* d_status = nullptr;
    TeamCtl* d_team = nullptr;     // one control block per member
    PanelTeamCtl* d_pteam = nullptr;   // cooperative panels: one control block per sorted position
    char* d_rings = nullptr;       // KB_TEAM_SLOTS records per member
    float stage_ms[KBDM_NSTAGES] = {0};
    bool timed = false;
    int mode = 0;                  // KBDM_MODE_* bits (kbdm_plan_set_mode)
    // pinned host staging of kbdm_plan_submit / kbdm_plan_collect (host -> host without a blocking copy)
    cd* h_signals = nullptr;
    char* h_out = nullptr;
    size_t h_out_bytes = 0;
    bool submitted = false;
};

namespace {

size_t item_arena_elems(int m, int l) {
    (void)l;
    // A, Q, P, R, H : m*m complex each; then, counted in complex units, the divide-and-conquer workspace (six real
    // m x m arrays)
    return 5 * (size_t)m * m + (size_t)(std::max(dc_ws_doubles(m), ab_ws_doubles(l > 0 ? l : m)) + 1) / 2;
}

int set_lds_attr() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_svd_fac), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dc_setup), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ab_iter), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ab_tail), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 2048));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hess), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wy_tfac), hipFuncAttributeMaxDynamicSharedMemorySize, KB_WY_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wy_apply), hipFuncAttributeMaxDynamicSharedMemorySize, KB_WY_APPLY_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bidiag_panel_team), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hess_panel_team), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hqr2), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hqr2_team), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_invit), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_invit_reg<8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    return KBDM_OK;
}

// Build items / order / chunks / device descriptors.  mode: 0 = full pipeline,
// 1 = stage API (dense m*m host layouts, hk_off).
int plan_build(kbdm_plan* pl, const int32_t* sig_idx, const int32_t* m, const int32_t* l, int nlanes = 1) {
    kbdm_ctx* ctx = pl->ctx;
    const int B = pl->B;
    pl->items.resize(B);
    pl->line_off.assign(B + 1, 0);
    pl->sv_off.assign(B + 1, 0);
    int64_t hk = 0;
    for (int i = 0; i < B; ++i) {
        KbItem& it = pl->items[i];
        memset(&it, 0, sizeof(it));
        it.m = m[i];
        it.l = l ? l[i] : m[i];
        it.sig = sig_idx ? sig_idx[i] : 0;
        it.q = pl->q;
        if (it.m < 1 || it.l < 1 || it.l > it.m) return fail(KBDM_E_INVALID, "item with invalid m/l");
        if (pl->N > 0 && (2 * it.m + pl->p - 1 > pl->N || it.sig < 0 || it.sig >= pl->S))
            return fail(KBDM_E_INVALID, "m can't be greater than (n + 1 - p)/2 / bad signal index");
        pl->line_off[i + 1] = pl->line_off[i] + it.l;
        pl->sv_off[i + 1] = pl->sv_off[i] + it.m;
        it.line_off = pl->line_off[i];
        it.sv_off = pl->sv_off[i];
        it.hk_off = hk;
        hk += (int64_t)it.m * it.m;
    }
    pl->total_lines = pl->line_off[B];
    pl->total_sv = pl->sv_off[B];
    // order by descending cost (~m^3): the longest-running workgroups start first
    pl->perm.resize(B);
    std::iota(pl->perm.begin(), pl->perm.end(), 0);
    std::stable_sort(pl->perm.begin(), pl->perm.end(),
                     [&](int a, int b) { return pl->items[a].m > pl->items[b].m; });
    // chunks: consecutive sorted items whose workspace fits the budget
    const size_t budget = (size_t)(ctx->ws_budget_gib * 1024.0 * 1024.0 * 1024.0) / sizeof(cd);
    pl->chunks.clear();
    size_t used = 0, vused = 0, dused = 0, acct = 0;
    Chunk cur;
    pl->arena_elems = 0;
    pl->varena_elems = 0;
    for (int pos = 0; pos < B; ++pos) {
        KbItem& it = pl->items[pl->perm[pos]];
        const size_t need = item_arena_elems(it.m, it.l);
        if (need > budget) return fail(KBDM_E_NOMEM, "one item exceeds the workspace budget");
        if (cur.count > 0 && acct + need > budget) {
            pl->chunks.push_back(cur);
            cur = Chunk();
            cur.first = pos;
            used = 0; vused = 0; dused = 0; acct = 0;
        }
        acct += need;
        it.dc_off = (long long)dused; dused += (size_t)std::max(dc_ws_doubles(it.m), ab_ws_doubles(it.l));
        pl->dc_elems = std::max(pl->dc_elems, dused);
        const size_t M = (size_t)it.m * it.m;
        size_t o = used;
        it.off[KB_BUF_A] = o; o += M;
        it.off[KB_BUF_Q] = o; o += M;
        it.off[KB_BUF_P] = o; o += M;
        it.off[KB_BUF_R] = o; o += M;
        it.off[KB_BUF_H] = o; o += M;
        used = o;
        it.vstride = (it.m + 1) & ~1;
        it.voff = (long long)vused;
        vused += (size_t)KB_V_SLOTS * it.vstride;
        cur.count++;
        cur.mmax = std::max(cur.mmax, it.m);
        cur.lmax = std::max(cur.lmax, it.l);
        pl->arena_elems = std::max(pl->arena_elems, used);
        pl->varena_elems = std::max(pl->varena_elems, vused);
    }
    if (cur.count > 0) pl->chunks.push_back(cur);
    for (size_t g = 0; g < pl->chunks.size(); ++g) pl->chunks[g].group = (int)g;
    // split every group into lanes of (about) equal cost, largest members in lane 0
    if (nlanes > 1) {
        std::vector<Chunk> out;
        for (const Chunk& g : pl->chunks) {
            int nl = std::min(nlanes, g.count / 4);
            if (nl < 2) { out.push_back(g); continue; }
            double total = 0.0;
            for (int k = 0; k < g.count; ++k) { const double mm = pl->items[pl->perm[g.first + k]].m; total += mm * mm * mm; }
            int pos = 0;
            double acc = 0.0;
            for (int ln = 0; ln < nl; ++ln) {
                Chunk c;
                c.group = g.group; c.lane = ln; c.first = g.first + pos;
                const double upto = (nl == 2) ? total * (ln == 0 ? ctx->lane0_frac : 1.0) : total * (ln + 1) / nl;
                while (pos < g.count && (ln == nl - 1 || acc < upto || c.count == 0)) {
                    const KbItem& it = pl->items[pl->perm[g.first + pos]];
                    acc += (double)it.m * it.m * it.m;
                    c.mmax = std::max(c.mmax, it.m);
                    c.lmax = std::max(c.lmax, it.l);
                    c.count++; pos++;
                }
                if (c.count > 0) out.push_back(c);
            }
        }
        pl->chunks.swap(out);
    }
    return KBDM_OK;
}

int plan_alloc(kbdm_plan* pl) {
    const int B = pl->B;
    HIPCHK(hipMalloc(&pl->d_items, sizeof(KbItem) * B));
    HIPCHK(hipMalloc(&pl->d_perm, sizeof(int) * B));
    HIPCHK(hipMemcpy(pl->d_items, pl->items.data(), sizeof(KbItem) * B, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pl->d_perm, pl->perm.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&pl->d_arena, sizeof(cd) * std::max<size_t>(pl->arena_elems, 1)));
    HIPCHK(hipMalloc(&pl->d_varena, sizeof(double) * std::max<size_t>(pl->varena_elems, 1)));
    HIPCHK(hipMalloc(&pl->d_lines, sizeof(double) * 4 * std::max<int64_t>(pl->total_lines, 1)));
    HIPCHK(hipMalloc(&pl->d_sv, sizeof(double) * std::max<int64_t>(pl->total_sv, 1)));
    HIPCHK(hipMalloc(&pl->d_mu, sizeof(cd) * std::max<int64_t>(pl->total_lines, 1)));
    HIPCHK(hipMalloc(&pl->d_keep, std::max<int64_t>(pl->total_lines, 1)));
    HIPCHK(hipMalloc(&pl->d_status, sizeof(int) * B));
    HIPCHK(hipMalloc(&pl->d_iwork, sizeof(int) * (4 * std::max(B, 1) + KB_QUEUE_WORDS)));
    HIPCHK(hipMemset(pl->d_iwork, 0, sizeof(int) * (4 * std::max(B, 1) + KB_QUEUE_WORDS)));
    HIPCHK(hipMalloc(&pl->d_team, sizeof(TeamCtl) * std::max(B, 1)));
    HIPCHK(hipMalloc(&pl->d_pteam, sizeof(PanelTeamCtl) * std::max(B, 1)));
    HIPCHK(hipMemset(pl->d_pteam, 0, sizeof(PanelTeamCtl) * std::max(B, 1)));
    HIPCHK(hipMalloc(&pl->d_rings, (size_t)std::max(B, 1) * KB_TEAM_SLOTS *
                                       team2_rec_bytes(pl->ctx->win_hqr2)));
    HIPCHK(hipMalloc(&pl->d_dc, sizeof(double) * std::max<size_t>(pl->dc_elems, 1)));
    HIPCHK(hipMalloc(&pl->d_abstat, sizeof(int) * 16 * KB_AB_BUDGET));
    HIPCHK(hipMemset(pl->d_abstat, 0, sizeof(int) * 16 * KB_AB_BUDGET));
    HIPCHK(hipMalloc(&pl->d_needqr, sizeof(int) * std::max(B, 1)));
    HIPCHK(hipMemset(pl->d_needqr, 0, sizeof(int) * std::max(B, 1)));
    if (pl->S > 0 && pl->N > 0) HIPCHK(hipMalloc(&pl->d_signals, sizeof(cd) * (size_t)pl->S * pl->N));
    return KBDM_OK;
}

struct StageTimer {
    kbdm_plan* pl;
    Chunk* ch;
    int idx = 0;
    int init() {
        if (ch->ev.empty()) {
            ch->ev.resize(KBDM_NSTAGES + 1);
            for (auto& e : ch->ev) HIPCHK(hipEventCreate(&e));
        }
        for (int k = 0; k < KBDM_NKCLASSES; ++k) { ch->kused[k] = 0; ch->klaunches[k] = 0; }
        HIPCHK(hipEventRecord(ch->ev[0], pl->ctx->lanes[ch->lane].stream));
        idx = 1;
        return KBDM_OK;
    }
    int mark() {
        HIPCHK(hipEventRecord(ch->ev[idx], pl->ctx->lanes[ch->lane].stream));
        idx++;
        return KBDM_OK;
    }
};

// Bracket `launches` launches of kernel class k on stream st with a pair of events (only in KBDM_MODE_KERNEL_TIMERS runs:
// the events cost host time and a little stream time, so the throughput runs do without).
struct KBracket {
    kbdm_plan* pl; Chunk* ch; int k; hipStream_t st; bool on;
    KBracket(kbdm_plan* pl_, Chunk& ch_, int k_, hipStream_t st_, int launches) : pl(pl_), ch(&ch_), k(k_), st(st_) {
        on = (pl->mode & KBDM_MODE_KERNEL_TIMERS) != 0;
        if (!on) return;
        mark();
        ch->klaunches[k] += launches;
    }
    ~KBracket() { if (on) mark(); }
    void mark() {
        auto& v = ch->kev[k];
        if ((int)v.size() <= ch->kused[k]) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } v.push_back(e); }
        hipEventRecord(v[ch->kused[k]++], st);
    }
};

int smem_fac(int n, int nt) { return KB_RED_BYTES + bidiag_scratch_bytes(n, nt / 64, 64); }

// Cooperative panels: workgroups per member of this chunk's panel launches.  A team's workgroups wait for each other, so a
// launch gets teams only when all its workgroups fit the context's budget of resident workgroups (and never in the
// conservative retry mode); results do not depend on the team size (kb_team.hpp).
int panel_team_size(const kbdm_plan* pl, const Chunk& ch) {
    const kbdm_ctx* ctx = pl->ctx;
    if (pl->mode & KBDM_MODE_SOLO_QR) return 1;
    if (ch.lane != 0 && !ctx->panel_T_all) return 1;
    int T = std::max(1, ctx->panel_T);
    // (teams on every lane: the lanes' launches run at the same time and share the budget)
    const int budget = ctx->panel_T_all ? ctx->panel_budget / std::max(1, ctx->nlanes) : ctx->panel_budget;
    while (T > 1 && (long long)((ch.count + 7) / 8 * 8) * T > budget) --T;
    return T;
}
// rows of the row-product partial sums that fit the LDS next to the other scratch (a multiple of 64; 0: nothing fits)
int panel_team_zr(int n, int nt) {
    const int fixed = KB_RED_BYTES + panel_team_scratch_bytes(n, 0, nt);
    int zr = (LDS_MAX - fixed) / (KB_TEAM_NCG * (int)sizeof(cd)) / 64 * 64;
    zr = std::min(zr, (n + 63) / 64 * 64);              // (all the rows in one batch when they fit)
    return zr;
}

// Explicit unitary factors of a chunk: members with n >= KB_WY_MIN by blocked compact-WY accumulation on FP64 MFMA
// (the T factors of all blocks in one launch, then one launch per block of 64 reflectors, last block first), the smaller
// ones by the per-column kernel k_gen
// (smallest register-chunk count that covers the largest of them).
int launch_gen(kbdm_plan* pl, Chunk& ch, int nmax, int mode, int nmat, hipStream_t gst) {
    kbdm_ctx* ctx = pl->ctx;
    const int* perm = pl->d_perm + ch.first;
    const int wy = ctx->gen_wy && nmax >= KB_WY_MIN;
    if (wy) {
        hipLaunchKernelGGL(k_wy_init, dim3(64, ch.count, nmat), dim3(256), 0, gst, pl->d_items, perm, pl->d_arena, pl->d_varena, mode);
        const int nblk = (nmax + KB_WYB - 1) / KB_WYB;
        hipLaunchKernelGGL(k_wy_tfac, dim3(nblk, ch.count, nmat), dim3(256), KB_WY_LDS, gst, pl->d_items, perm, pl->d_arena, pl->d_varena,
                           mode);
        for (int s = 0; s < nblk; ++s) {
            // a member's trailing matrix at step s has at most (s + 1) 64 + 2 rows (its last block is the partial one)
            const int span = std::min(nmax, (s + 2) * KB_WYB);
            KBracket kb(pl, ch, KBDM_K_WY_APPLY, gst, 1);
            hipLaunchKernelGGL(k_wy_apply, dim3((span + 63) / 64, ch.count, nmat), dim3(256), KB_WY_APPLY_LDS, gst, pl->d_items, perm,
                               pl->d_arena, pl->d_varena, mode, s);
        }
    }
    // the per-column kernel for the members below KB_WY_MIN (all of them when the blocked path is off)
    int nsmall = 0;
    for (int i = 0; i < ch.count; ++i) {
        const KbItem& it = pl->items[pl->perm[ch.first + i]];
        const int n = mode == 0 ? it.m : it.l;
        if (!wy || n < KB_WY_MIN) nsmall = std::max(nsmall, n);
    }
    if (nsmall > 0) {
        dim3 grid(ctx->split_gen, ch.count, nmat), block(256);
        const int chunks = (nsmall + 63) / 64;
        if (chunks <= 2) hipLaunchKernelGGL(k_gen<2>, grid, block, 0, gst, pl->d_items, perm, pl->d_arena, pl->d_varena, mode, wy);
        else if (chunks <= 4) hipLaunchKernelGGL(k_gen<4>, grid, block, 0, gst, pl->d_items, perm, pl->d_arena, pl->d_varena, mode, wy);
        else if (chunks <= 8) hipLaunchKernelGGL(k_gen<8>, grid, block, 0, gst, pl->d_items, perm, pl->d_arena, pl->d_varena, mode, wy);
        else if (chunks <= 16) hipLaunchKernelGGL(k_gen<16>, grid, block, 0, gst, pl->d_items, perm, pl->d_arena, pl->d_varena, mode, wy);
        else if (chunks <= 32) hipLaunchKernelGGL(k_gen<32>, grid, block, 0, gst, pl->d_items, perm, pl->d_arena, pl->d_varena, mode, wy);
        else return fail(KBDM_E_NOMEM, "m larger than 2048 is not supported by k_gen");
    }
    return KBDM_OK;
}

// Bidiagonal SVD by divide and conquer (kb_bdsdc.hpp, kbdm_dc_kernels.hpp): the tree needs only (d, e) and runs on the
// side stream while the main stream accumulates Q and P; then L = Q X and R = P Y as real GEMMs.
int launch_svd_dc(kbdm_plan* pl, Chunk& ch, StageTimer* tm) {
    kbdm_ctx* ctx = pl->ctx;
    Lane& ln = ctx->lanes[ch.lane];
    hipStream_t st = ln.stream;
    hipStream_t ds = ln.stream2;                    // (a lane without a side stream runs the tree in stream order)
    const int* perm = pl->d_perm + ch.first;
    if (ds != st) {
        HIPCHK(hipEventRecord(ln.ev_fork, st));
        HIPCHK(hipStreamWaitEvent(ds, ln.ev_fork, 0));
    }
    // per step: the deepest level any member still has, and the largest node of that step
    const int Lmax = dc_depth(ch.mmax);
    std::vector<int> nmax(Lmax, 0), dmax(Lmax, -1);
    {
        int last_m = -1;
        for (int i = 0; i < ch.count; ++i) {
            const int m = pl->items[pl->perm[ch.first + i]].m;
            if (m == last_m) continue;               // (sorted by m: equal sizes are adjacent)
            last_m = m;
            const int L = dc_depth(m);
            for (int s = 0; s < L; ++s) {
                const int depth = L - 1 - s;
                int n = m;
                for (int k = 0; k < depth; ++k) n = n - 1 - (n - 1) / 2;      // the larger child
                nmax[s] = std::max(nmax[s], n);
                dmax[s] = std::max(dmax[s], depth);
            }
        }
    }
    {
        const int sm = KB_RED_BYTES + dc_leaf_scratch_bytes(KB_DC_LEAF);
        hipLaunchKernelGGL(k_dc_leaf, dim3(1 << Lmax, ch.count), dim3(128), sm, ds, pl->d_items, perm, pl->d_varena, pl->d_dc, sm);
    }
    for (int s = 0; s < Lmax; ++s) {
        const int sm = KB_RED_BYTES + dc_merge_scratch_bytes(nmax[s]);
        if (sm > LDS_MAX - 64) return fail(KBDM_E_NOMEM, "m too large for the divide-and-conquer scratch");
        const int nt = std::min(1024, std::max(64, (nmax[s] + 63) / 64 * 64));
        hipLaunchKernelGGL(k_dc_setup, dim3(1 << dmax[s], ch.count), dim3(nt), sm, ds, pl->d_items, perm, pl->d_varena,
                           pl->d_dc, s, pl->d_status, sm);
        const int tmax = (nmax[s] + 1 + 63) / 64;
        hipLaunchKernelGGL(k_dc_apply, dim3((1 << dmax[s]) * tmax * tmax, 2, ch.count), dim3(256), 0, ds, pl->d_items, perm,
                           pl->d_dc, s, tmax);
    }
    if (ds != st) HIPCHK(hipEventRecord(ln.ev_join, ds));
    {
        int r = launch_gen(pl, ch, ch.mmax, 0, 2, st);
        if (r) return r;
        if (tm) { r = tm->mark(); if (r) return r; }
        if (ds != st) HIPCHK(hipStreamWaitEvent(st, ln.ev_join, 0));
        if (tm) { r = tm->mark(); if (r) return r; }      // k_dc_tree slot (overlapped with k_gen(Q,P))
    }
    hipLaunchKernelGGL(k_dc_final, dim3((2 * ch.mmax + 63) / 64, (ch.mmax + 63) / 64, 2 * ch.count), dim3(256), 0, st,
                       pl->d_items, perm, pl->d_arena, pl->d_dc);
    if (tm) { int r = tm->mark(); if (r) return r; }
    hipLaunchKernelGGL(k_dc_sv, dim3(ch.count), dim3(256), KB_RED_BYTES, st, pl->d_items, perm, pl->d_varena, pl->d_dc, pl->d_sv);
    if (tm) { int r = tm->mark(); if (r) return r; }
    HIPCHK(hipGetLastError());
    return KBDM_OK;
}

int launch_svd(kbdm_plan* pl, Chunk& ch, StageTimer* tm) {
    kbdm_ctx* ctx = pl->ctx;
    Lane& ln = ctx->lanes[ch.lane];
    hipStream_t st = ln.stream;
    const int* perm = pl->d_perm + ch.first;
    {
        // blocked part: panels + MFMA trailing updates (all inside the "k_svd_fac" stage timer)
        const int npan = ctx->blocked ? bidiag_num_panels(ch.mmax) : 0;
        const int T = panel_team_size(pl, ch);
        const int zr = panel_team_zr(ch.mmax, ctx->nt_fac);
        const int smt = KB_RED_BYTES + panel_team_scratch_bytes(ch.mmax, zr, ctx->nt_fac);
        if (npan > 0) {
            if (zr < 64) return fail(KBDM_E_NOMEM, "m too large for the panel scratch");
            HIPCHK(hipMemsetAsync(pl->d_pteam + ch.first, 0, sizeof(PanelTeamCtl) * ch.count, st));
        }
        for (int pnl = 0; pnl < npan; ++pnl) {
            {
                KBracket kb(pl, ch, KBDM_K_BIDIAG_PANEL, st, 1);
                hipLaunchKernelGGL(k_bidiag_panel_team, dim3((ch.count + 7) / 8 * 8 * T), dim3(ctx->nt_fac), smt, st, pl->d_items,
                                   perm, pl->d_arena, pl->d_varena, pnl, smt, T, ch.count, pl->d_pteam + ch.first, zr,
                                   pl->d_status);
            }
            const int nn = ch.mmax - (pnl + 1) * KB_NB;
            const int tiles = (nn + 63) / 64;
            {
                KBracket kb(pl, ch, KBDM_K_TRAIL_UPDATE, st, 1);
                hipLaunchKernelGGL(k_trail_update, dim3(tiles, tiles, ch.count), dim3(256), 0, st, pl->d_items, perm,
                                   pl->d_arena, pnl);
            }
        }
        int sm = smem_fac(ch.mmax, ctx->nt_fac);
        if (sm > LDS_MAX) return fail(KBDM_E_NOMEM, "m too large for the bidiagonalisation scratch");
        // room for the block behind the panels (fewer than NB + NX columns), which k_svd_fac then reduces in LDS
        if (npan > 0) sm = std::max(sm, std::min(LDS_MAX, KB_RED_BYTES + bidiag_tail_lds_bytes(KB_NB + KB_NX - 1, ctx->nt_fac / 64, 64)));
        hipLaunchKernelGGL(k_svd_fac, dim3(ch.count), dim3(ctx->nt_fac), sm, st, pl->d_items, perm, pl->d_arena,
                           pl->d_varena, sm, npan > 0 ? 1 : 0);
        if (tm) { int r = tm->mark(); if (r) return r; }
    }
    return launch_svd_dc(pl, ch, tm);
}

int launch_eig(kbdm_plan* pl, Chunk& ch, StageTimer* tm) {
    kbdm_ctx* ctx = pl->ctx;
    Lane& ln = ctx->lanes[ch.lane];
    hipStream_t st = ln.stream;
    const int* perm = pl->d_perm + ch.first;
    {
        const int npan = ctx->blocked ? bidiag_num_panels(ch.lmax) : 0;
        const int T = panel_team_size(pl, ch);
        const int zr = panel_team_zr(ch.lmax, ctx->nt_fac);
        const int smt = KB_RED_BYTES + panel_team_scratch_bytes(ch.lmax, zr, ctx->nt_fac);
        if (npan > 0) {
            if (zr < 64) return fail(KBDM_E_NOMEM, "l too large for the Hessenberg panel scratch");
            HIPCHK(hipMemsetAsync(pl->d_pteam + ch.first, 0, sizeof(PanelTeamCtl) * ch.count, st));
        }
        for (int pnl = 0; pnl < npan; ++pnl) {
            {
                KBracket kb(pl, ch, KBDM_K_HESS_PANEL, st, 1);
                hipLaunchKernelGGL(k_hess_panel_team, dim3((ch.count + 7) / 8 * 8 * T), dim3(ctx->nt_fac), smt, st, pl->d_items,
                                   perm, pl->d_arena, pl->d_varena, pnl, smt, T, ch.count, pl->d_pteam + ch.first, zr,
                                   pl->d_status);
            }
            const int ncol = ch.lmax - (pnl + 1) * KB_NB;
            {
                KBracket kb(pl, ch, KBDM_K_HESS_Z, st, 1);
                const int nzt = (ncol + 63) / 64, nyt = (pnl * KB_NB + 64) / 64;     // Z's tiles + the tiles of the rows above the panel
                hipLaunchKernelGGL(k_hess_z, dim3(nzt + nyt, ch.count), dim3(256), 0, st, pl->d_items, perm, pl->d_arena, pnl, nzt);
            }
            {
                KBracket kb(pl, ch, KBDM_K_HESS_UPDATE, st, 1);
                hipLaunchKernelGGL(k_hess_update, dim3((ch.lmax + 63) / 64, (ncol + 63) / 64, ch.count), dim3(256), 0, st,
                                   pl->d_items, perm, pl->d_arena, pnl);
            }
        }
        const int sm = KB_RED_BYTES + gehd2_scratch_bytes(ch.lmax, ctx->nt_fac / 64, 64);
        if (sm > LDS_MAX) return fail(KBDM_E_NOMEM, "l too large for the Hessenberg scratch");
        hipLaunchKernelGGL(k_hess, dim3(ch.count), dim3(ctx->nt_fac), sm, st, pl->d_items, perm, pl->d_arena,
                           pl->d_varena, sm, npan > 0 ? 1 : 0);
        if (tm) { int r = tm->mark(); if (r) return r; }
    }
    // Qh is first needed by k_gemm<3>: accumulate it on a side stream while the QR iteration runs.  A lane
    // without a side stream of its own borrows the critical lane's, which is idle during the QR iteration
    // (a stream of this lane's own would cost a hardware queue; in stream order k_gen(Qh) would sit in front
    // of the QR iteration: 6.6 ms of the non-critical lane's chain on C2).
    hipStream_t qs = (ln.stream2 != ln.stream) ? ln.stream2 : ctx->lanes[0].stream2;
    HIPCHK(hipEventRecord(ln.ev_fork, st));
    HIPCHK(hipStreamWaitEvent(qs, ln.ev_fork, 0));
    {
        int r = launch_gen(pl, ch, ch.lmax, 1, 1, qs);
        if (r) return r;
        HIPCHK(hipEventRecord(ln.ev_join, qs));
        if (tm) { r = tm->mark(); if (r) return r; }      // k_gen(Qh) slot (overlapped with k_hqr)
    }
    {
        const int win = ctx->win_hqr2;
        const int sm = KB_RED_BYTES + hqr2_scratch_bytes(win);
        MsStats* prof = nullptr;
        const bool do_prof = ctx->hqr_prof != 0;
        if (do_prof) {
            HIPCHK(hipMalloc(&prof, sizeof(MsStats) * pl->B));
            HIPCHK(hipMemsetAsync(prof, 0, sizeof(MsStats) * pl->B, st));
        }
        // Fast path: divide-and-conquer Ehrlich-Aberth for every member (kb_aberth.hpp): leaves, then KB_AB_BUDGET
        // iteration launches per level (a tile whose roots have settled only copies them through), the check; the QR
        // iteration below then runs for the members the path declined only.  The conservative retry mode skips it.
        const bool aberth = ctx->eig_ab && !(pl->mode & KBDM_MODE_SOLO_QR);
        if (aberth) {
            const int Dmax = ab_depth(ch.lmax);
            int lmin_tail = ch.lmax, ltail = 1;                // smallest member: does any qualify for the tail kernel?  largest that does
            for (int i = 0; i < ch.count; ++i) {
                const int l = pl->items[pl->perm[ch.first + i]].l;
                lmin_tail = std::min(lmin_tail, l);
                if (l <= KB_AB_TAIL_MAXL) ltail = std::max(ltail, l);
            }
            int Dmin = Dmax;                                   // steps before Dmin - 1 are inner levels for every member
            std::vector<int> gx(Dmax, 0);                      // workgroups per member and step: tiles x nodes of its level
            {
                int last_l = -1;
                for (int i = 0; i < ch.count; ++i) {
                    const int l = pl->items[pl->perm[ch.first + i]].l;
                    if (l == last_l) continue;
                    last_l = l;
                    const int D = ab_depth(l);
                    Dmin = std::min(Dmin, D);
                    for (int s2 = 0; s2 < D; ++s2) {
                        const int depth = D - 1 - s2;
                        const int Tl = (ab_level_nmax(l, depth) + KB_AB_TILE - 1) / KB_AB_TILE;
                        gx[s2] = std::max(gx[s2], Tl << depth);
                    }
                }
            }
            const int sml = KB_RED_BYTES;
            hipLaunchKernelGGL(k_ab_leaf, dim3(1 << Dmax, ch.count), dim3(64), sml, st, pl->d_items, perm, pl->d_arena, pl->d_varena,
                               pl->d_dc, pl->d_needqr, sml);
            {
                int nlaunch = 0;
                for (int s2 = 0; s2 < Dmax; ++s2) nlaunch += (s2 < Dmin - 1 ? KB_AB_INNER_BUDGET : KB_AB_BUDGET);
                KBracket kb(pl, ch, KBDM_K_AB_ITER, st, nlaunch);
                for (int s2 = 0; s2 < Dmax; ++s2)
                    for (int itn = 0; itn < (s2 < Dmin - 1 ? KB_AB_INNER_BUDGET : KB_AB_BUDGET); ++itn) {
                        hipLaunchKernelGGL(k_ab_iter, dim3(gx[s2], ch.count), dim3(256), sizeof(AbLds), st, pl->d_items, perm, pl->d_arena,
                                           pl->d_varena, pl->d_dc, pl->d_needqr, s2, itn, pl->d_abstat,
                                           (ctx->ab_dbg & ~16) | (ctx->ab_tail ? 0 : 16));
                        // the root level's tail: a wavefront per root for members with few unsettled roots left (a member is
                        // taken by exactly one of the two kernels in an iteration: kb_aberth.hpp, ab_tail_takes)
                        if (ctx->ab_tail && s2 >= Dmin - 1 && itn >= KB_AB_TAIL_FROM && lmin_tail <= KB_AB_TAIL_MAXL)
                            hipLaunchKernelGGL(k_ab_tail, dim3(KB_AB_TAIL_WGS, ch.count), dim3(256), ab_tail_lds_bytes(ltail), st, pl->d_items,
                                               perm, pl->d_arena, pl->d_varena, pl->d_dc, pl->d_needqr, s2, itn, pl->d_abstat,
                                               ab_tail_npad(ltail));
                    }
            }
            hipLaunchKernelGGL(k_ab_finish, dim3(ch.count), dim3(256), KB_RED_BYTES, st, pl->d_items, perm, pl->d_arena, pl->d_varena,
                               pl->d_dc, pl->d_mu, pl->d_needqr);
        }
        const int* needqr = aberth ? pl->d_needqr : nullptr;
        // Large members of the critical lane run as two-workgroup teams (k_hqr_team); the remaining
        // members of the chunk run solo on the side stream (after k_gen(Qh)), concurrently.
        int nteam = 0;
        if (!aberth && ctx->team_hqr && !(pl->mode & KBDM_MODE_SOLO_QR) && win > 0 && ln.stream2 != ln.stream &&
            (ch.lane == 0 || ctx->team_hqr > 1)) {
            int nside = 0;                                   // lanes that may run teams share the budget of resident teams
            for (int i = 0; i < ctx->nlanes; ++i) nside += (ctx->lanes[i].stream2 != ctx->lanes[i].stream) ? 1 : 0;
            const int cap = std::max(1, ctx->team_max / std::max(1, nside));
            while (nteam < ch.count && nteam < cap && pl->items[pl->perm[ch.first + nteam]].l >= ctx->team_min_l) ++nteam;
        }
        if (nteam > 0) {
            hipLaunchKernelGGL(k_hqr2_team, dim3(2 * nteam), dim3(512), sm, st, pl->d_items, perm, pl->d_arena,
                               pl->d_mu, pl->d_status, sm, ctx->nb_hqr2, win, pl->d_team, pl->d_rings, prof);
        }
        if (ch.count > nteam) {
            hipStream_t ss = nteam > 0 ? ln.stream2 : st;     // (the fallback behind the Aberth path stays on the main stream)
            // a bounded number of workgroups takes the members from a queue (they are sorted by size,
            // largest first): the launch lasts as long as its largest member either way, and the CUs it
            // does not occupy go to the other lane's / the next ensemble's throughput-bound stages
            const int nsolo = ch.count - nteam;
            const int cidx = (int)(&ch - pl->chunks.data());
            int nwg = ctx->hqr_wgs;
            if (nwg < 0) {
                // the iteration costs ~l^2 per member: sum l^2 / lmax^2 workgroups keep pace with the largest
                // member (15 % slack for the greedy packing); a large batch gets one workgroup per member
                double work = 0.0, lmax = 1.0;
                for (int i = nteam; i < ch.count; ++i) {
                    const double l = pl->items[pl->perm[ch.first + i]].l;
                    work += l * l;
                    lmax = std::max(lmax, l);
                }
                nwg = (int)std::ceil(1.15 * work / (lmax * lmax));
                if (nwg > 192) nwg = 0;
                if (aberth) nwg = std::min(std::max(nwg, 1), 16);      // a handful of workgroups for the rare fallback
            }
            int* queue = (nwg > 0 && nsolo > nwg && cidx < KB_QUEUE_WORDS) ? pl->d_iwork + 4 * pl->B + cidx : nullptr;
            hipLaunchKernelGGL(k_hqr2, dim3(queue ? nwg : nsolo), dim3(512), sm, ss, pl->d_items, perm + nteam,
                               pl->d_arena, pl->d_mu, pl->d_status, sm, ctx->nb_hqr2, win, prof, nsolo, queue, needqr);
            if (nteam > 0) HIPCHK(hipEventRecord(ln.ev_join, ln.stream2));   // the join now covers Qh and the solo members
        }
        if (do_prof) {   // diagnostic build path only: synchronous dump of the largest item's counters
            std::vector<MsStats> h(pl->B);
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipMemcpy(h.data(), prof, sizeof(MsStats) * pl->B, hipMemcpyDeviceToHost));
            const MsStats& x = h[pl->perm[ch.first]];
            fprintf(stderr, "[k_hqr prof] n=%d batches=%lld intervals=%lld winsteps=%lld singles=%lld | Mcycles: total=%.1f scan=%.1f shift=%.1f load=%.1f chase=%.1f store=%.1f strip=%.1f single=%.1f | wave0 tiles=%.1f tload=%.1f treplay=%.1f tstore=%.1f\n",
                    pl->items[pl->perm[ch.first]].l, x.batches, x.intervals, x.small_steps, x.single_sweeps, x.cyc_total / 1e6,
                    x.cyc_scan / 1e6, x.cyc_shift / 1e6, x.cyc_load / 1e6, x.cyc_chase / 1e6, x.cyc_store / 1e6,
                    x.cyc_strip / 1e6, x.cyc_single / 1e6, x.ntiles / 1e6, x.cyc_tload / 1e6, x.cyc_treplay / 1e6, x.cyc_tstore / 1e6);
            hipFree(prof);
        }
        if (tm) { int r = tm->mark(); if (r) return r; }
    }
    {
        HIPCHK(hipStreamWaitEvent(st, ln.ev_join, 0));   // eigenvalues of the solo members (side stream) and Qh
        // One launch per SIZE CLASS of the members of the chunk (a member is solved by the same kernel instantiation in
        // every batch): l <= 128, <= 256, <= 512 register-resident with the multipliers in LDS; <= 1280 the streaming
        // form (four wavefronts per workgroup = one per SIMD: a wavefront may then use accumulation registers next to
        // its 256 VGPRs); beyond that (or with KBDM_INVIT_REG=0) the LDS-resident form.
        int lmaxc[5] = {0, 0, 0, 0, 0};                   // largest l per class
        int cfirst[5] = {0, 0, 0, 0, 0}, clast[5] = {-1, -1, -1, -1, -1};   // range of the chunk that holds the class
        for (int i = 0; i < ch.count; ++i) {
            const int l = pl->items[pl->perm[ch.first + i]].l;
            const int c = !ctx->invit_reg ? 4 : (l <= 128 ? 0 : l <= 256 ? 1 : l <= 512 ? 2 : l <= KB_INVIT_BIG_MAXC * 64 ? 3 : 4);
            if (!lmaxc[c]) cfirst[c] = i;
            clast[c] = i;
            lmaxc[c] = std::max(lmaxc[c], l);
        }
        auto nsplit = [](int lmax, int wpb) { return std::max(1, std::min((lmax + 2 * wpb - 1) / (2 * wpb), 64)); };
        // wavefronts per workgroup of the register-resident form (a solve does not depend on it: one eigenvalue per wavefront)
        const int w2 = ctx->invit_wpb, w4 = ctx->invit_wpb, w8 = ctx->invit_wpb8;
        if (lmaxc[0]) hipLaunchKernelGGL(k_invit_reg<2>, dim3(clast[0] - cfirst[0] + 1, nsplit(lmaxc[0], w2)), dim3(64 * w2), w2 * 2 * 64 * sizeof(cd), st, pl->d_items, perm + cfirst[0], pl->d_arena, pl->d_varena, pl->d_mu, pl->d_status, 0);
        if (lmaxc[1]) hipLaunchKernelGGL(k_invit_reg<4>, dim3(clast[1] - cfirst[1] + 1, nsplit(lmaxc[1], w4)), dim3(64 * w4), w4 * 4 * 64 * sizeof(cd), st, pl->d_items, perm + cfirst[1], pl->d_arena, pl->d_varena, pl->d_mu, pl->d_status, 128);
        if (lmaxc[2]) hipLaunchKernelGGL(k_invit_reg<8>, dim3(clast[2] - cfirst[2] + 1, nsplit(lmaxc[2], w8)), dim3(64 * w8), w8 * 8 * 64 * sizeof(cd), st, pl->d_items, perm + cfirst[2], pl->d_arena, pl->d_varena, pl->d_mu, pl->d_status, 256);
        if (lmaxc[3]) hipLaunchKernelGGL(k_invit_big<KB_INVIT_BIG_MAXC>, dim3(clast[3] - cfirst[3] + 1, nsplit(lmaxc[3], 4)), dim3(256), 0, st, pl->d_items, perm + cfirst[3], pl->d_arena, pl->d_varena, pl->d_mu, pl->d_status, 512);
        if (lmaxc[4]) {
            const int per4 = invit_scratch_bytes_per_wave(lmaxc[4]);
            int nw4 = std::min(ctx->nt_invit / 64, (LDS_MAX - 64 - KB_RED_BYTES) / per4);
            if (nw4 < 1) return fail(KBDM_E_NOMEM, "l too large for the inverse-iteration scratch");
            const int sm4 = KB_RED_BYTES + nw4 * per4;
            hipLaunchKernelGGL(k_invit, dim3(clast[4] - cfirst[4] + 1, ctx->split_invit), dim3(ctx->nt_invit), sm4, st, pl->d_items, perm + cfirst[4],
                               pl->d_arena, pl->d_varena, pl->d_mu, pl->d_status, sm4, ctx->invit_reg ? KB_INVIT_BIG_MAXC * 64 : 0);
        }
        HIPCHK(hipStreamWaitEvent(st, ln.ev_join, 0));   // Qh ready before k_gemm<3>
        if (tm) { int r = tm->mark(); if (r) return r; }
    }
    HIPCHK(hipGetLastError());
    return KBDM_OK;
}

template <int STAGE>
void launch_gemm(kbdm_plan* pl, Chunk& ch, int Mmax, int Nmax) {
    const int* perm = pl->d_perm + ch.first;
    dim3 grid((Mmax + GT - 1) / GT, (Nmax + GT - 1) / GT, ch.count);
    hipLaunchKernelGGL(k_gemm<STAGE>, grid, dim3(256), 0, pl->ctx->lanes[ch.lane].stream, pl->d_items, perm, pl->d_signals, pl->N,
                       pl->p, pl->d_arena, pl->d_varena);
}

// The whole pipeline of one chunk on its lane's streams.
int run_chunk(kbdm_plan* pl, Chunk& ch) {
    hipStream_t st = pl->ctx->lanes[ch.lane].stream;
    StageTimer tm{pl, &ch};
    int r = tm.init();
    if (r) return r;
    const int* perm = pl->d_perm + ch.first;
    {   // K1: U^{p-1} into the SVD work buffer
        HankelOut o0{pl->d_arena, pl->p - 1, KB_BUF_A, 1};
        HankelOut none{nullptr, 0, 0, 0};
        const int tiles = (ch.mmax + HK_TILE - 1) / HK_TILE;
        {
            KBracket kb(pl, ch, KBDM_K_HANKEL, st, 1);
            hipLaunchKernelGGL(k_hankel, dim3(tiles, tiles, ch.count), dim3(256), 0, st, pl->d_items, perm,
                               pl->d_signals, pl->N, 1, o0, none, none);
        }
        if ((r = tm.mark())) return r;
    }
    if ((r = launch_svd(pl, ch, &tm))) return r;
    launch_gemm<1>(pl, ch, ch.mmax, ch.lmax);
    if ((r = tm.mark())) return r;
    launch_gemm<2>(pl, ch, ch.lmax, ch.lmax);
    if ((r = tm.mark())) return r;
    if ((r = launch_eig(pl, ch, &tm))) return r;
    launch_gemm<3>(pl, ch, ch.lmax, ch.lmax);
    if ((r = tm.mark())) return r;
    launch_gemm<4>(pl, ch, ch.mmax, ch.lmax);
    if ((r = tm.mark())) return r;
    launch_gemm<5>(pl, ch, ch.mmax, ch.lmax);
    if ((r = tm.mark())) return r;
    {
        const int wpb = 4;
        dim3 grid((ch.lmax + wpb - 1) / wpb, ch.count);
        hipLaunchKernelGGL(k_epilogue, grid, dim3(64 * wpb), 0, st, pl->d_items, perm, pl->d_signals, pl->N,
                           pl->d_arena, pl->d_mu, pl->dwell, pl->d_lines, pl->d_keep);
        if ((r = tm.mark())) return r;
    }
    HIPCHK(hipGetLastError());
    return KBDM_OK;
}

// Trailer of a packed block: the sender's rank and the sequence number of the gather it belongs to.  The ranks must issue
// the gathers of all their contexts in the same order (they share one communicator); a slip would deliver a block of ANOTHER
// step without any error from the transport - the receiver's check turns it into an error instead.
constexpr unsigned KB_TRAILER_MAGIC = 0x4B42444Du;        // "KBDM"
struct KbTrailer { unsigned magic, rank; unsigned long long seq; };
__global__ void k_write_trailer(char* block_end, unsigned rank, unsigned long long seq) {
    KbTrailer t{KB_TRAILER_MAGIC, rank, seq};
    *reinterpret_cast<KbTrailer*>(block_end - sizeof(KbTrailer)) = t;
}
constexpr int KB_TRAILER_MAX_WORLD = 64;
struct KbBlockOffs { long long o[KB_TRAILER_MAX_WORLD + 1]; };
__global__ void k_check_trailers(const char* gathered, KbBlockOffs offs, int world, unsigned long long seq, int* bad) {
    const long long* off = offs.o;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= world || off[r + 1] == off[r]) return;
    const KbTrailer t = *reinterpret_cast<const KbTrailer*>(gathered + off[r + 1] - sizeof(KbTrailer));
    if (t.magic != KB_TRAILER_MAGIC || t.rank != (unsigned)r || t.seq != seq) atomicOr(bad, 1);
}
std::atomic<unsigned long long> g_gather_seq{0};           // one communicator per process: one sequence

}  // namespace

extern "C" {

int kbdm_abi_version(void) { return KBDM_ABI_VERSION; }

const char* kbdm_last_error(void) { return g_err.c_str(); }

int kbdm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* kbdm_stage_name(int stage) {
    return (stage >= 0 && stage < KBDM_NSTAGES) ? kStageNames[stage] : "";
}

int kbdm_ctx_create(int device, kbdm_ctx** out) { return kbdm_ctx_create_lanes(device, 0, out); }

int kbdm_ctx_create_lanes(int device, int nlanes, kbdm_ctx** out) {
    if (!out) return fail(KBDM_E_INVALID, "null out pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(KBDM_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(KBDM_E_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    kbdm_ctx* c = new kbdm_ctx();
    c->device = device;
    c->nlanes = std::min(KB_MAX_LANES, std::max(1, nlanes > 0 ? nlanes : env_int("KBDM_LANES", c->nlanes)));
    const int hwq = std::max(1, env_int("GPU_MAX_HW_QUEUES", 4));
    const bool side_all = env_int("KBDM_SIDE_ALL", 0) != 0;
    if (const char* v = getenv("KBDM_LANE0_FRAC")) c->lane0_frac = std::min(0.95, std::max(0.05, atof(v)));
    for (int i = 0; i < c->nlanes; ++i) {
        Lane& ln = c->lanes[i];
        HIPCHK(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        // Side streams only while every stream still has a hardware queue of its own (GPU_MAX_HW_QUEUES,
        // 4 by default): streams beyond that share queues, and two lanes whose replay kernels wait
        // (in-kernel flags) on generator kernels queued behind each other would never finish.  A lane
        // without a side stream runs its side work in stream order (and its QR iteration solo).
        // (Measured: side streams / teams on the other lanes as well slow the critical lane down more than
        // they speed those lanes up - C2 182 ms instead of 175 - so they are opt-in: KBDM_SIDE_ALL=1.)
        if (i == 0 || (side_all && c->nlanes + i + 1 <= hwq)) {
            HIPCHK(hipStreamCreateWithFlags(&ln.stream2, hipStreamNonBlocking));
            } else ln.stream2 = ln.stream;
        HIPCHK(hipEventCreateWithFlags(&ln.ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ln.ev_join, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ln.ev_done, hipEventDisableTiming));
    }
    c->stream = c->lanes[0].stream;
    HIPCHK(hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming));
    c->nt_fac = env_int("KBDM_NT_FAC", c->nt_fac);
    c->nt_invit = env_int("KBDM_NT_INVIT", c->nt_invit);
    c->blocked = env_int("KBDM_BLOCKED", c->blocked);
    c->gen_wy = env_int("KBDM_GEN_WY", c->gen_wy);
    c->eig_ab = env_int("KBDM_EIG_AB", c->eig_ab);
    c->hqr_prof = env_int("KBDM_HQR_PROF", c->hqr_prof);
    c->panel_T = std::min(32, std::max(1, env_int("KBDM_PANEL_T", c->panel_T)));
    c->panel_budget = std::max(8, env_int("KBDM_PANEL_BUDGET", c->panel_budget));
    c->panel_T_all = env_int("KBDM_PANEL_T_ALL", c->panel_T_all);
    c->ab_dbg = env_int("KBDM_AB_DBG", 0);
    c->ab_tail = env_int("KBDM_AB_TAIL", c->ab_tail);
#if !defined(KBDM_DEBUG_BUILD)
    c->ab_dbg &= ~(2 | 4);
#endif
    c->nb_hqr2 = std::min(KB2_NBMAX, std::max(1, env_int("KBDM_NB_HQR2", c->nb_hqr2)));
    c->win_hqr2 = KB2_WIN_DEV;                       // the device chase is compiled for this window
    c->split_gen = std::max(1, env_int("KBDM_SPLIT_GEN", c->split_gen));
    c->split_invit = std::max(1, env_int("KBDM_SPLIT_INVIT", c->split_invit));
    c->team_hqr = env_int("KBDM_TEAM_HQR", c->team_hqr);
    c->invit_reg = env_int("KBDM_INVIT_REG", c->invit_reg);
    c->invit_wpb = std::min(8, std::max(1, env_int("KBDM_INVIT_WPB", c->invit_wpb)));
    c->invit_wpb8 = std::min(8, std::max(1, env_int("KBDM_INVIT_WPB8", c->invit_wpb8)));
    c->team_min_l = env_int("KBDM_TEAM_MIN_L", c->team_min_l);
    c->hqr_wgs = std::max(-1, env_int("KBDM_HQR_WGS", c->hqr_wgs));
    c->team_max = std::min(120, std::max(1, env_int("KBDM_TEAM_MAX", c->team_max)));
    if (const char* v = getenv("KBDM_WS_GIB")) c->ws_budget_gib = atof(v);
    int r = set_lds_attr();
    if (r) { delete c; return r; }
    *out = c;
    return KBDM_OK;
}

int kbdm_ctx_set_panel_teams(kbdm_ctx* ctx, int T, int budget, int all_lanes, double lane0_frac) {
    if (!ctx || T < 1 || T > 32 || budget < 8) return fail(KBDM_E_INVALID, "bad panel-team policy");
    ctx->panel_T = T;
    ctx->panel_budget = budget;
    ctx->panel_T_all = all_lanes ? 1 : 0;
    if (lane0_frac > 0.0) ctx->lane0_frac = std::min(0.95, std::max(0.05, lane0_frac));
    return KBDM_OK;
}

int kbdm_ctx_destroy(kbdm_ctx* ctx) {
    if (!ctx) return KBDM_OK;
    hipSetDevice(ctx->device);
    if (ctx->comm) kbdm_comm_destroy(ctx);          // (synchronises the streams it used: before any of them is destroyed)
    if (ctx->comm_stream) { hipStreamSynchronize(ctx->comm_stream); hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
    for (int i = 0; i < KB_MAX_LANES; ++i) {
        Lane& ln = ctx->lanes[i];
        if (ln.stream2 && ln.stream2 != ln.stream) hipStreamDestroy(ln.stream2);
        if (ln.stream) hipStreamDestroy(ln.stream);
        ln.stream = ln.stream2 = nullptr;
        if (ln.ev_fork) hipEventDestroy(ln.ev_fork);
        if (ln.ev_join) hipEventDestroy(ln.ev_join);
        if (ln.ev_done) hipEventDestroy(ln.ev_done);
    }
    ctx->stream = nullptr;
    if (ctx->ev_start) hipEventDestroy(ctx->ev_start);
    if (ctx->ev_packed) hipEventDestroy(ctx->ev_packed);
    if (ctx->ev_gathered) hipEventDestroy(ctx->ev_gathered);
    if (ctx->d_pack) hipFree(ctx->d_pack);
    if (ctx->d_gather) hipFree(ctx->d_gather);
    if (ctx->d_gather_bad) hipFree(ctx->d_gather_bad);
    delete ctx;
    return KBDM_OK;
}

int kbdm_plan_create(kbdm_ctx* ctx, int S, int N, int B, const int32_t* sig_idx, const int32_t* m,
                     const int32_t* l, int p, double q, double dwell, kbdm_plan** out) {
    if (!ctx || !out || !m || B < 0 || S < 1 || N < 1 || p < 1) return fail(KBDM_E_INVALID, "bad plan arguments");
    HIPCHK(hipSetDevice(ctx->device));
    kbdm_plan* pl = new kbdm_plan();
    pl->ctx = ctx; pl->S = S; pl->N = N; pl->B = B; pl->p = p; pl->q = q; pl->dwell = dwell;
    int r = plan_build(pl, sig_idx, m, l, ctx->nlanes);
    if (!r) r = plan_alloc(pl);
    if (r) { kbdm_plan_destroy(pl); return r; }
    *out = pl;
    return KBDM_OK;
}

int kbdm_plan_destroy(kbdm_plan* pl) {
    if (!pl) return KBDM_OK;
    hipFree(pl->d_signals); hipFree(pl->d_items); hipFree(pl->d_perm); hipFree(pl->d_arena);
    hipFree(pl->d_varena); hipFree(pl->d_lines); hipFree(pl->d_sv); hipFree(pl->d_mu);
    hipFree(pl->d_keep); hipFree(pl->d_status); hipFree(pl->d_iwork);
    hipFree(pl->d_team); hipFree(pl->d_pteam); hipFree(pl->d_rings); hipFree(pl->d_dc); hipFree(pl->d_needqr); hipFree(pl->d_abstat);
    if (pl->h_signals) hipHostFree(pl->h_signals);
    if (pl->h_out) hipHostFree(pl->h_out);
    for (auto& ch : pl->chunks)
        for (auto& e : ch.ev) hipEventDestroy(e);
    delete pl;
    return KBDM_OK;
}

int64_t kbdm_plan_total_lines(const kbdm_plan* pl) { return pl ? pl->total_lines : 0; }
int64_t kbdm_plan_total_sv(const kbdm_plan* pl) { return pl ? pl->total_sv : 0; }

int kbdm_plan_offsets(const kbdm_plan* pl, int64_t* line_off, int64_t* sv_off) {
    if (!pl) return fail(KBDM_E_INVALID, "null plan");
    if (line_off) memcpy(line_off, pl->line_off.data(), sizeof(int64_t) * (pl->B + 1));
    if (sv_off) memcpy(sv_off, pl->sv_off.data(), sizeof(int64_t) * (pl->B + 1));
    return KBDM_OK;
}

int kbdm_plan_upload(kbdm_plan* pl, const double* signals_host) {
    if (!pl || !signals_host) return fail(KBDM_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipMemcpyAsync(pl->d_signals, signals_host, sizeof(cd) * (size_t)pl->S * pl->N, hipMemcpyHostToDevice,
                          pl->ctx->stream));
    HIPCHK(hipStreamSynchronize(pl->ctx->stream));
    return KBDM_OK;
}

int kbdm_plan_execute(kbdm_plan* pl) {
    if (!pl) return fail(KBDM_E_INVALID, "null plan");
    if (pl->B == 0) return KBDM_OK;
    kbdm_ctx* ctx = pl->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemsetAsync(pl->d_status, 0, sizeof(int) * pl->B, st));
    HIPCHK(hipMemsetAsync(pl->d_iwork, 0, sizeof(int) * (4 * pl->B + KB_QUEUE_WORDS), st));
    HIPCHK(hipMemsetAsync(pl->d_team, 0, sizeof(TeamCtl) * pl->B, st));
    HIPCHK(hipMemsetAsync(pl->d_needqr, 0, sizeof(int) * pl->B, st));
    // groups one after the other (they share the arena); inside a group one chunk per lane, all
    // lanes concurrently: fork from the main stream, join back into it
    size_t ci = 0;
    while (ci < pl->chunks.size()) {
        size_t ce = ci;
        while (ce < pl->chunks.size() && pl->chunks[ce].group == pl->chunks[ci].group) ++ce;
        HIPCHK(hipEventRecord(ctx->ev_start, st));
        for (size_t c = ci; c < ce; ++c) {
            Chunk& ch = pl->chunks[c];
            hipStream_t ls = ctx->lanes[ch.lane].stream;
            if (ch.lane != 0) HIPCHK(hipStreamWaitEvent(ls, ctx->ev_start, 0));
            int r = run_chunk(pl, ch);
            if (r) return r;
            if (ch.lane != 0) HIPCHK(hipEventRecord(ctx->lanes[ch.lane].ev_done, ls));
        }
        for (size_t c = ci; c < ce; ++c)
            if (pl->chunks[c].lane != 0) HIPCHK(hipStreamWaitEvent(st, ctx->lanes[pl->chunks[c].lane].ev_done, 0));
        ci = ce;
    }
    pl->timed = true;
    return KBDM_OK;
}

int kbdm_plan_sync(kbdm_plan* pl) {
    if (!pl) return fail(KBDM_E_INVALID, "null plan");
    HIPCHK(hipStreamSynchronize(pl->ctx->stream));
    return KBDM_OK;
}

int kbdm_plan_wait_stage(kbdm_plan* pl, int stage) {
    if (!pl || stage < 0 || stage >= KBDM_NSTAGES) return fail(KBDM_E_INVALID, "bad stage");
    if (!pl->timed) return KBDM_OK;
    for (auto& ch : pl->chunks)
        if (ch.lane == 0 && !ch.ev.empty()) {
            HIPCHK(hipEventSynchronize(ch.ev[stage + 1]));
            break;
        }
    return KBDM_OK;
}

int kbdm_plan_stage_ms(kbdm_plan* pl, float* ms, int n) {
    if (!pl || !ms) return fail(KBDM_E_INVALID, "null argument");
    HIPCHK(hipStreamSynchronize(pl->ctx->stream));
    for (int s = 0; s < KBDM_NSTAGES; ++s) pl->stage_ms[s] = 0.f;
    if (pl->timed)
        for (auto& ch : pl->chunks) {
            if (ch.ev.empty() || ch.lane != 0) continue;       // lane 0 holds the largest members: the critical path
            for (int s = 0; s < KBDM_NSTAGES; ++s) {
                float t = 0.f;
                HIPCHK(hipEventElapsedTime(&t, ch.ev[s], ch.ev[s + 1]));
                pl->stage_ms[s] += t;
            }
        }
    for (int s = 0; s < n && s < KBDM_NSTAGES; ++s) ms[s] = pl->stage_ms[s];
    return KBDM_OK;
}

const char* kbdm_kernel_class_name(int k) { return (k >= 0 && k < KBDM_NKCLASSES) ? kKernelClassNames[k] : nullptr; }

int kbdm_plan_kernel_ms(kbdm_plan* pl, int k, float* total_ms, int32_t* launches) {
    if (!pl || !total_ms || !launches || k < 0 || k >= KBDM_NKCLASSES) return fail(KBDM_E_INVALID, "bad kernel class");
    for (int i = 0; i < pl->ctx->nlanes; ++i)
        if (pl->ctx->lanes[i].stream) HIPCHK(hipStreamSynchronize(pl->ctx->lanes[i].stream));
    if (pl->ctx->lanes[0].stream2) HIPCHK(hipStreamSynchronize(pl->ctx->lanes[0].stream2));
    float tot = 0.f;
    int n = 0;
    for (auto& ch : pl->chunks) {
        if (ch.lane != 0) continue;                 // lane 0 holds the largest members: the launches the stage timers cover
        for (int e = 0; e + 1 < ch.kused[k]; e += 2) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, ch.kev[k][e], ch.kev[k][e + 1]));
            tot += t;
        }
        n += ch.klaunches[k];
    }
    *total_ms = tot;
    *launches = n;
    return KBDM_OK;
}

int kbdm_plan_eig_fallbacks(kbdm_plan* pl) {
    if (!pl || pl->B == 0) return 0;
    std::vector<int> h(pl->B);
    if (hipStreamSynchronize(pl->ctx->stream) != hipSuccess ||
        hipMemcpy(h.data(), pl->d_needqr, sizeof(int) * pl->B, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int n = 0;
    for (int v : h) n += v ? 1 : 0;
    return n;
}

int kbdm_plan_ab_stats(kbdm_plan* pl, int32_t* out, int n) {
    if (!pl || !out) return fail(KBDM_E_INVALID, "null argument");
    std::vector<int> h(16 * KB_AB_BUDGET);
    HIPCHK(hipStreamSynchronize(pl->ctx->stream));
    HIPCHK(hipMemcpy(h.data(), pl->d_abstat, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < (int)h.size(); ++i) out[i] = h[i];
    HIPCHK(hipMemset(pl->d_abstat, 0, sizeof(int) * h.size()));
    return KBDM_OK;
}

int kbdm_plan_lane0_members(const kbdm_plan* pl) {
    if (!pl) return 0;
    int n = 0;
    for (const auto& ch : pl->chunks)
        if (ch.lane == 0) n += ch.count;
    return n;
}

int kbdm_plan_download(kbdm_plan* pl, double* lines, double* sv, double* mu, uint8_t* keep, int32_t* status) {
    if (!pl) return fail(KBDM_E_INVALID, "null plan");
    hipStream_t st = pl->ctx->stream;
    HIPCHK(hipSetDevice(pl->ctx->device));
    if (lines && pl->total_lines) HIPCHK(hipMemcpyAsync(lines, pl->d_lines, sizeof(double) * 4 * pl->total_lines, hipMemcpyDeviceToHost, st));
    if (sv && pl->total_sv) HIPCHK(hipMemcpyAsync(sv, pl->d_sv, sizeof(double) * pl->total_sv, hipMemcpyDeviceToHost, st));
    if (mu && pl->total_lines) HIPCHK(hipMemcpyAsync(mu, pl->d_mu, sizeof(cd) * pl->total_lines, hipMemcpyDeviceToHost, st));
    if (keep && pl->total_lines) HIPCHK(hipMemcpyAsync(keep, pl->d_keep, pl->total_lines, hipMemcpyDeviceToHost, st));
    if (status && pl->B) HIPCHK(hipMemcpyAsync(status, pl->d_status, sizeof(int) * pl->B, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (status) {
        const int force = g_force_status.load();                      // test hook: pretend the members failed
        if (force) for (int i = 0; i < pl->B; ++i) status[i] |= force;
    }
    return KBDM_OK;
}

int kbdm_debug_force_status(int always, int once) {
    g_force_status.store(always);
    g_force_status_once.store(once);
    return KBDM_OK;
}

#if defined(KB_PANEL_PROF)
// diagnostic builds only (tools/panel_phases.py): device address of a __device__ symbol of this library
int kbdm_debug_symbol(const char* name, void** ptr, size_t* bytes) {
    if (strcmp(name, "kb_panel_prof") != 0) return KBDM_E_INVALID;
    HIPCHK(hipGetSymbolAddress(ptr, HIP_SYMBOL(kb_panel_prof)));
    HIPCHK(hipGetSymbolSize(bytes, HIP_SYMBOL(kb_panel_prof)));
    return KBDM_OK;
}
#endif

int kbdm_plan_set_mode(kbdm_plan* pl, int mode) {
    if (!pl || (mode & ~(KBDM_MODE_SOLO_QR | KBDM_MODE_KERNEL_TIMERS))) return fail(KBDM_E_INVALID, "bad mode");
    pl->mode = mode;
    return KBDM_OK;
}

int64_t kbdm_workspace_estimate(int B, const int32_t* m, const int32_t* l) {
    if (B < 0 || (B > 0 && !m)) return -1;
    // what plan_alloc asks hipMalloc for, before chunking: the matrix arena + rotation log dominate
    double tot = 0.0;
    for (int i = 0; i < B; ++i) {
        const int li = l ? l[i] : m[i];
        tot += (double)item_arena_elems(m[i], li) * sizeof(cd) + 9.0 * 8.0 * (m[i] + 2) + 64.0 * li + 8.0 * m[i] +
               (double)KB_TEAM_SLOTS * 4096.0 + 512.0;
    }
    return (int64_t)tot;
}

int64_t kbdm_plan_workspace_bytes(const kbdm_plan* pl) {
    if (!pl) return 0;
    return (int64_t)(sizeof(cd) * pl->arena_elems + sizeof(double) * pl->varena_elems + sizeof(double) * pl->dc_elems + 57 * (size_t)pl->total_lines + 8 * (size_t)pl->total_sv +
                     (size_t)pl->B * (sizeof(KbItem) + sizeof(TeamCtl) + 24) + sizeof(cd) * (size_t)pl->S * pl->N);
}

namespace {
// layout of the pinned output block of kbdm_plan_submit: lines | sv | mu | status | keep
struct OutLayout { size_t lines, sv, mu, status, keep, total; };
OutLayout out_layout(const kbdm_plan* pl) {
    OutLayout o;
    o.lines = 0;
    o.sv = o.lines + 32 * (size_t)pl->total_lines;
    o.mu = o.sv + 8 * (size_t)pl->total_sv;
    o.status = o.mu + 16 * (size_t)pl->total_lines;
    o.keep = o.status + 4 * (size_t)pl->B;
    o.total = ((o.keep + (size_t)pl->total_lines + 63) & ~(size_t)63) + 64;
    return o;
}
}  // namespace

int kbdm_plan_submit(kbdm_plan* pl, const double* signals_host) {
    if (!pl) return fail(KBDM_E_INVALID, "null plan");
    hipStream_t st = pl->ctx->stream;
    HIPCHK(hipSetDevice(pl->ctx->device));
    const OutLayout o = out_layout(pl);
    const size_t sig_bytes = sizeof(cd) * (size_t)pl->S * pl->N;
    if (!pl->h_out) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&pl->h_out), o.total, hipHostMallocDefault));
        pl->h_out_bytes = o.total;
        if (sig_bytes) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&pl->h_signals), sig_bytes, hipHostMallocDefault));
    }
    if (pl->submitted) HIPCHK(hipStreamSynchronize(st));      // the staging buffers are still the previous run's
    if (signals_host) {
        if (!pl->d_signals) return fail(KBDM_E_INVALID, "the plan has no signals");
        memcpy(pl->h_signals, signals_host, sig_bytes);
        HIPCHK(hipMemcpyAsync(pl->d_signals, pl->h_signals, sig_bytes, hipMemcpyHostToDevice, st));
    }
    int r = kbdm_plan_execute(pl);
    if (r) return r;
    char* h = pl->h_out;
    if (pl->total_lines) {
        HIPCHK(hipMemcpyAsync(h + o.lines, pl->d_lines, 32 * (size_t)pl->total_lines, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h + o.mu, pl->d_mu, 16 * (size_t)pl->total_lines, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h + o.keep, pl->d_keep, (size_t)pl->total_lines, hipMemcpyDeviceToHost, st));
    }
    if (pl->total_sv) HIPCHK(hipMemcpyAsync(h + o.sv, pl->d_sv, 8 * (size_t)pl->total_sv, hipMemcpyDeviceToHost, st));
    if (pl->B) HIPCHK(hipMemcpyAsync(h + o.status, pl->d_status, 4 * (size_t)pl->B, hipMemcpyDeviceToHost, st));
    pl->submitted = true;
    return KBDM_OK;
}

int kbdm_plan_collect(kbdm_plan* pl, double* lines, double* sv, double* mu, uint8_t* keep, int32_t* status) {
    if (!pl) return fail(KBDM_E_INVALID, "null plan");
    if (!pl->submitted) return fail(KBDM_E_INVALID, "kbdm_plan_collect without kbdm_plan_submit");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipStreamSynchronize(pl->ctx->stream));
    pl->submitted = false;
    const OutLayout o = out_layout(pl);
    const char* h = pl->h_out;
    if (lines) memcpy(lines, h + o.lines, 32 * (size_t)pl->total_lines);
    if (sv) memcpy(sv, h + o.sv, 8 * (size_t)pl->total_sv);
    if (mu) memcpy(mu, h + o.mu, 16 * (size_t)pl->total_lines);
    if (keep) memcpy(keep, h + o.keep, (size_t)pl->total_lines);
    if (status) {
        memcpy(status, h + o.status, 4 * (size_t)pl->B);
        const int force = g_force_status.load() | g_force_status_once.exchange(0);   // test hooks: pretend the members failed
        if (force) for (int i = 0; i < pl->B; ++i) status[i] |= force;
    }
    return KBDM_OK;
}

void* kbdm_plan_lines_device(kbdm_plan* pl) { return pl ? pl->d_lines : nullptr; }
void* kbdm_plan_sv_device(kbdm_plan* pl) { return pl ? pl->d_sv : nullptr; }

int kbdm_plan_copy_lines_device(kbdm_plan* pl, void* dst, int64_t dst_bytes) {
    if (!pl || !dst) return fail(KBDM_E_INVALID, "null argument");
    const int64_t need = (int64_t)sizeof(double) * 4 * pl->total_lines;
    if (dst_bytes < need) return fail(KBDM_E_INVALID, "destination too small");
    HIPCHK(hipSetDevice(pl->ctx->device));
    HIPCHK(hipMemcpyAsync(dst, pl->d_lines, need, hipMemcpyDeviceToDevice, pl->ctx->stream));
    HIPCHK(hipStreamSynchronize(pl->ctx->stream));
    return KBDM_OK;
}

// ---- RCCL, bound at run time (the library loads and solves without it; only the gather of a sharded ensemble needs it)
namespace {
struct KbUid { char b[KBDM_UNIQUE_ID_BYTES]; };      // ncclUniqueId (passed by value)
struct RcclApi {
    void* so = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, KbUid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
constexpr int kNcclUint8 = 1;

int rccl_load() {
    if (g_rccl.so) return KBDM_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* so = nullptr;
    for (const char* n : names)
        if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!so) return fail(KBDM_E_HIP, std::string("librccl.so not found: ") + dlerror());
    RcclApi a;
    a.so = so;
    *(void**)&a.GetUniqueId = dlsym(so, "ncclGetUniqueId");
    *(void**)&a.CommInitRank = dlsym(so, "ncclCommInitRank");
    *(void**)&a.CommDestroy = dlsym(so, "ncclCommDestroy");
    *(void**)&a.GroupStart = dlsym(so, "ncclGroupStart");
    *(void**)&a.GroupEnd = dlsym(so, "ncclGroupEnd");
    *(void**)&a.Send = dlsym(so, "ncclSend");
    *(void**)&a.Recv = dlsym(so, "ncclRecv");
    *(void**)&a.GetErrorString = dlsym(so, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv)
        return fail(KBDM_E_HIP, "librccl.so lacks an expected symbol");
    g_rccl = a;
    return KBDM_OK;
}
#define NCCLCHK(expr)                                                                                       \
    do {                                                                                                    \
        int e__ = (expr);                                                                                   \
        if (e__ != 0)                                                                                       \
            return fail(KBDM_E_HIP, std::string(#expr) + ": " +                                             \
                                        (g_rccl.GetErrorString ? g_rccl.GetErrorString(e__) : "rccl error")); \
    } while (0)
}  // namespace

int kbdm_comm_unique_id(unsigned char* id_out) {
    if (!id_out) return fail(KBDM_E_INVALID, "null argument");
    int r = rccl_load();
    if (r) return r;
    NCCLCHK(g_rccl.GetUniqueId(id_out));
    return KBDM_OK;
}

int kbdm_comm_init(kbdm_ctx* ctx, int world, int rank, const unsigned char* id) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail(KBDM_E_INVALID, "bad communicator arguments");
    if (ctx->comm) return fail(KBDM_E_INVALID, "the context already owns a communicator");
    int r = rccl_load();
    if (r) return r;
    HIPCHK(hipSetDevice(ctx->device));
    KbUid uid;
    memcpy(uid.b, id, KBDM_UNIQUE_ID_BYTES);
    NCCLCHK(g_rccl.CommInitRank(&ctx->comm, world, uid, rank));
    ctx->comm_owned = true;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return KBDM_OK;
}

int kbdm_comm_attach(kbdm_ctx* ctx, kbdm_ctx* owner) {
    if (!ctx || !owner || !owner->comm || ctx == owner) return fail(KBDM_E_INVALID, "bad attach arguments");
    if (ctx->comm) return fail(KBDM_E_INVALID, "the context already has a communicator");
    if (ctx->device != owner->device) return fail(KBDM_E_INVALID, "contexts on different devices cannot share a communicator");
    ctx->comm = owner->comm;
    ctx->comm_owned = false;
    ctx->comm_world = owner->comm_world;
    ctx->comm_rank = owner->comm_rank;
    return KBDM_OK;
}

int kbdm_comm_destroy(kbdm_ctx* ctx) {
    if (!ctx || !ctx->comm) return KBDM_OK;
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) hipStreamSynchronize(ctx->comm_stream);
    if (ctx->comm_owned) g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_owned = false;
    ctx->comm_world = 0;
    return KBDM_OK;
}

int64_t kbdm_packed_bytes(int64_t lines, int64_t sv, int64_t members) {
    const int64_t raw = 32 * lines + 8 * sv + 4 * members + lines;
    return ((raw + 15) & ~(int64_t)15) + 16;        // + the trailer {magic, rank, sequence number of the gather}
}

void* kbdm_gathered_device(kbdm_ctx* ctx) { return ctx ? ctx->d_gather : nullptr; }

int kbdm_plan_gather(kbdm_plan* pl, int world, int rank, const int64_t* bytes, int root, void* host_out) {
    if (!pl || !bytes || world < 1 || rank < 0 || rank >= world || root >= world) return fail(KBDM_E_INVALID, "bad gather arguments");
    kbdm_ctx* ctx = pl->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t mine = kbdm_packed_bytes(pl->total_lines, pl->total_sv, pl->B);
    if (bytes[rank] != mine) return fail(KBDM_E_INVALID, "bytes[rank] does not match this plan's packed size");
    if ((world > 1 || ctx->comm) && (!ctx->comm || ctx->comm_world != world || ctx->comm_rank != rank))
        return fail(KBDM_E_INVALID, "no communicator for this world size / rank: call kbdm_comm_init first");
    int64_t total = 0;
    std::vector<int64_t> off(world + 1, 0);
    for (int r = 0; r < world; ++r) {
        if (bytes[r] < 0 || (bytes[r] & 15)) return fail(KBDM_E_INVALID, "block sizes must be multiples of 16");
        off[r + 1] = off[r] + bytes[r];
    }
    total = off[world];
    if ((size_t)mine > ctx->pack_cap) {
        if (ctx->d_pack) HIPCHK(hipFree(ctx->d_pack));
        ctx->d_pack = nullptr; ctx->pack_cap = 0;
        HIPCHK(hipMalloc(&ctx->d_pack, std::max<size_t>((size_t)mine, 16)));
        ctx->pack_cap = std::max<size_t>((size_t)mine, 16);
    }
    if ((size_t)total > ctx->gather_cap) {
        if (ctx->d_gather) HIPCHK(hipFree(ctx->d_gather));
        ctx->d_gather = nullptr; ctx->gather_cap = 0;
        HIPCHK(hipMalloc(&ctx->d_gather, std::max<size_t>((size_t)total, 16)));
        ctx->gather_cap = std::max<size_t>((size_t)total, 16);
    }
    // Pack on the plan's stream (ordered after the run it belongs to); the transfer itself runs on the context's
    // communication stream behind an event, so that neither the host nor the plan's stream waits for the collective's
    // kernel to find room on a busy GPU.  The pack buffer is reused only after the previous transfer has finished.
    if (!ctx->comm_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    hipStream_t cs = ctx->comm_stream;
    if (ctx->gather_pending) HIPCHK(hipStreamWaitEvent(st, ctx->ev_gathered, 0));
    char* d = ctx->d_pack;
    if (mine >= 32) HIPCHK(hipMemsetAsync(d + (mine - 32), 0, 16, st));   // the padding bytes are defined
    const unsigned long long seq = ++g_gather_seq;
    hipLaunchKernelGGL(k_write_trailer, dim3(1), dim3(1), 0, st, d + mine, (unsigned)rank, seq);
    if (!ctx->d_gather_bad) {
        HIPCHK(hipMalloc(&ctx->d_gather_bad, sizeof(int)));
        HIPCHK(hipMemset(ctx->d_gather_bad, 0, sizeof(int)));
    }

    if (pl->total_lines) HIPCHK(hipMemcpyAsync(d, pl->d_lines, 32 * pl->total_lines, hipMemcpyDeviceToDevice, st));
    d += 32 * pl->total_lines;
    if (pl->total_sv) HIPCHK(hipMemcpyAsync(d, pl->d_sv, 8 * pl->total_sv, hipMemcpyDeviceToDevice, st));
    d += 8 * pl->total_sv;
    if (pl->B) HIPCHK(hipMemcpyAsync(d, pl->d_status, 4 * (size_t)pl->B, hipMemcpyDeviceToDevice, st));
    d += 4 * (size_t)pl->B;
    if (pl->total_lines) HIPCHK(hipMemcpyAsync(d, pl->d_keep, pl->total_lines, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipEventRecord(ctx->ev_packed, st));
    HIPCHK(hipStreamWaitEvent(cs, ctx->ev_packed, 0));
    const bool receive = root < 0 || root == rank;
    if (world == 1 && !ctx->comm) {
        if (mine) HIPCHK(hipMemcpyAsync(ctx->d_gather, ctx->d_pack, mine, hipMemcpyDeviceToDevice, cs));
    } else {
        // ONE grouped operation: every block travels once, straight between device buffers
        NCCLCHK(g_rccl.GroupStart());
        int err = 0;                              // the group is closed on every path out of here
        for (int r = 0; r < world && !err; ++r) {
            const bool to_r = root < 0 || r == root;
            if (to_r && mine) err = g_rccl.Send(ctx->d_pack, (size_t)mine, kNcclUint8, r, ctx->comm, cs);
            if (!err && receive && bytes[r])
                err = g_rccl.Recv(ctx->d_gather + off[r], (size_t)bytes[r], kNcclUint8, r, ctx->comm, cs);
        }
        const int end = g_rccl.GroupEnd();
        NCCLCHK(err);
        NCCLCHK(end);
    }
    if (receive && total && world <= KB_TRAILER_MAX_WORLD) {
        // every block that arrived carries its sender's rank and THIS gather's sequence number
        KbBlockOffs offs;
        for (int r = 0; r <= world; ++r) offs.o[r] = off[r];
        hipLaunchKernelGGL(k_check_trailers, dim3(1), dim3(KB_TRAILER_MAX_WORLD), 0, cs, ctx->d_gather, offs, world, seq, ctx->d_gather_bad);
    }
    if (host_out && receive && total) HIPCHK(hipMemcpyAsync(host_out, ctx->d_gather, total, hipMemcpyDeviceToHost, cs));
    HIPCHK(hipEventRecord(ctx->ev_gathered, cs));
    ctx->gather_pending = true;
    if (host_out) return kbdm_gather_wait(ctx);              // a host copy was asked for: the call completes it
    return KBDM_OK;
}

int kbdm_gather_wait(kbdm_ctx* ctx) {
    if (!ctx) return fail(KBDM_E_INVALID, "null context");
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->comm_stream || !ctx->gather_pending) return KBDM_OK;
    // Bounded: a rank that has died leaves its peers inside the grouped transfer for good.  Poll the event that closes the
    // gather and give up after KBDM_GATHER_TIMEOUT_S (default 300 s): the caller gets an error, the process exits with a
    // non-zero code and the launcher (llckbdm_amd.launch.spawn) tears the other ranks down.
    const double limit = std::max(1, env_int("KBDM_GATHER_TIMEOUT_S", 300));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ctx->ev_gathered);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); return fail(KBDM_E_HIP, "the gather failed on the device"); }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            return fail(KBDM_E_HIP, "kbdm_gather_wait: the gather did not finish within KBDM_GATHER_TIMEOUT_S (a peer rank is gone?)");
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    (void)hipGetLastError();
    if (ctx->d_gather_bad) {
        int bad = 0;
        HIPCHK(hipMemcpy(&bad, ctx->d_gather_bad, sizeof(int), hipMemcpyDeviceToHost));
        if (bad) {
            HIPCHK(hipMemset(ctx->d_gather_bad, 0, sizeof(int)));
            return fail(KBDM_E_HIP, "gather: a received block carries another step's sequence number or another rank's id "
                                    "(the ranks did not issue their gathers in the same order)");
        }
    }
    return KBDM_OK;
}

int kbdm_solve_batch(kbdm_ctx* ctx, const double* signals, int S, int N, int B, const int32_t* sig_idx,
                     const int32_t* m, const int32_t* l, int p, double q, double dwell, double* lines,
                     double* sv, double* mu, uint8_t* keep, int32_t* status) {
    kbdm_plan* pl = nullptr;
    int r = kbdm_plan_create(ctx, S, N, B, sig_idx, m, l, p, q, dwell, &pl);
    if (r) return r;
    r = kbdm_plan_upload(pl, signals);
    if (!r) r = kbdm_plan_execute(pl);
    if (!r) r = kbdm_plan_download(pl, lines, sv, mu, keep, status);
    kbdm_plan_destroy(pl);
    return r;
}

// ---------------------------------------------------------------- stage entry points
int kbdm_hankel_batch(kbdm_ctx* ctx, const double* signals, int S, int N, int B, const int32_t* sig_idx,
                      const int32_t* m, int p, double* U0, double* Up1, double* Up) {
    if (!ctx || !signals || !m) return fail(KBDM_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    kbdm_plan pl;
    pl.ctx = ctx; pl.S = S; pl.N = N; pl.B = B; pl.p = p;
    // Hankel only needs indices up to 2m-2+p <= N-1
    int r = plan_build(&pl, sig_idx, m, nullptr);
    if (r) return r;
    size_t tot = 0;
    int mmax = 0;
    for (auto& it : pl.items) { tot += (size_t)it.m * it.m; mmax = std::max(mmax, it.m); }
    cd *d_sig = nullptr, *d_out[3] = {nullptr, nullptr, nullptr};
    KbItem* d_items = nullptr;
    double* host_out[3] = {U0, Up1, Up};
    const int shifts[3] = {0, p - 1, p};
    DevBufs bufs;                                   // freed on every return below
    HIPCHK(bufs.alloc(&d_sig, sizeof(cd) * (size_t)S * N));
    HIPCHK(bufs.alloc(&d_items, sizeof(KbItem) * std::max(B, 1)));
    HIPCHK(hipMemcpy(d_sig, signals, sizeof(cd) * (size_t)S * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_items, pl.items.data(), sizeof(KbItem) * B, hipMemcpyHostToDevice));
    HankelOut o[3];
    int nout = 0;
    int which[3];
    for (int k = 0; k < 3; ++k)
        if (host_out[k]) {
            HIPCHK(bufs.alloc(&d_out[k], sizeof(cd) * std::max<size_t>(tot, 1)));
            o[nout] = HankelOut{d_out[k], shifts[k], 0, 0};
            which[nout++] = k;
        }
    for (int k = nout; k < 3; ++k) o[k] = HankelOut{nullptr, 0, 0, 0};
    if (nout > 0 && B > 0) {
        const int tiles = (mmax + HK_TILE - 1) / HK_TILE;
        hipLaunchKernelGGL(k_hankel, dim3(tiles, tiles, B), dim3(256), 0, ctx->stream, d_items, (const int*)nullptr,
                           d_sig, N, nout, o[0], o[1], o[2]);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < nout; ++k)
            HIPCHK(hipMemcpy(host_out[which[k]], d_out[which[k]], sizeof(cd) * tot, hipMemcpyDeviceToHost));
    }
    return KBDM_OK;
}

static int stage_plan(kbdm_ctx* ctx, int B, const int32_t* n, kbdm_plan** out) {
    kbdm_plan* pl = new kbdm_plan();
    pl->ctx = ctx; pl->S = 0; pl->N = 0; pl->B = B; pl->p = 1;
    int r = plan_build(pl, nullptr, n, nullptr);
    if (!r) r = plan_alloc(pl);
    if (r) { kbdm_plan_destroy(pl); return r; }
    *out = pl;
    return KBDM_OK;
}

int kbdm_svd_batch(kbdm_ctx* ctx, const double* A, int B, const int32_t* m, double* L, double* s, double* R,
                   int32_t* status) {
    if (!ctx || !A || !m) return fail(KBDM_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    kbdm_plan* pl = nullptr;
    int r = stage_plan(ctx, B, m, &pl);
    if (r) return r;
    size_t tot = 0;
    for (auto& it : pl->items) tot += (size_t)it.m * it.m;
    cd* d_dense = nullptr;
    hipStream_t st = ctx->stream;
    r = KBDM_OK;
    do {
        HIPTRY(hipMalloc(&d_dense, sizeof(cd) * std::max<size_t>(tot, 1)));
        HIPTRY(hipMemcpy(d_dense, A, sizeof(cd) * tot, hipMemcpyHostToDevice));
        HIPTRY(hipMemsetAsync(pl->d_status, 0, sizeof(int) * B, st));
        HIPTRY(hipMemsetAsync(pl->d_iwork, 0, sizeof(int) * (4 * B + KB_QUEUE_WORDS), st));
        for (auto& ch : pl->chunks) {
            // stage plans are built in one chunk by construction of the tests; handle generally
            hipLaunchKernelGGL(k_transpose_in, dim3(64, B), dim3(256), 0, st, pl->d_items, d_dense, pl->d_arena, KB_BUF_A, 0);
            if ((r = launch_svd(pl, ch, nullptr))) break;
        }
        if (r) break;
        HIPTRY(hipStreamSynchronize(st));
        if (L) {
            hipLaunchKernelGGL(k_transpose_out, dim3(64, B), dim3(256), 0, st, pl->d_items, pl->d_arena, KB_BUF_A, d_dense, 0);
            HIPTRY(hipStreamSynchronize(st));
            HIPTRY(hipMemcpy(L, d_dense, sizeof(cd) * tot, hipMemcpyDeviceToHost));
        }
        if (R) {
            hipLaunchKernelGGL(k_transpose_out, dim3(64, B), dim3(256), 0, st, pl->d_items, pl->d_arena, KB_BUF_R, d_dense, 0);
            HIPTRY(hipStreamSynchronize(st));
            HIPTRY(hipMemcpy(R, d_dense, sizeof(cd) * tot, hipMemcpyDeviceToHost));
        }
        if (s) HIPTRY(hipMemcpy(s, pl->d_sv, sizeof(double) * pl->total_sv, hipMemcpyDeviceToHost));
        if (status) HIPTRY(hipMemcpy(status, pl->d_status, sizeof(int) * B, hipMemcpyDeviceToHost));
        HIPTRY(hipGetLastError());
    } while (0);
    hipFree(d_dense);
    if (pl->chunks.size() > 1) r = fail(KBDM_E_NOMEM, "stage API batch exceeds the workspace budget");
    kbdm_plan_destroy(pl);
    return r;
}

__global__ void k_fill_ones(const KbItem* items, double* varena) {
    const KbItem it = items[blockIdx.x];
    for (int i = threadIdx.x; i < it.l; i += blockDim.x) varena[it.voff + KB_V_DSQI * it.vstride + i] = 1.0;
}

int kbdm_eig_batch(kbdm_ctx* ctx, const double* W, int B, const int32_t* n, double* mu, double* P, int32_t* status) {
    if (!ctx || !W || !n) return fail(KBDM_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    kbdm_plan* pl = nullptr;
    int r = stage_plan(ctx, B, n, &pl);
    if (r) return r;
    size_t tot = 0;
    for (auto& it : pl->items) tot += (size_t)it.m * it.m;
    cd* d_dense = nullptr;
    hipStream_t st = ctx->stream;
    do {
        if (pl->chunks.size() > 1) { r = fail(KBDM_E_NOMEM, "stage API batch exceeds the workspace budget"); break; }
        HIPTRY(hipMalloc(&d_dense, sizeof(cd) * std::max<size_t>(tot, 1)));
        HIPTRY(hipMemcpy(d_dense, W, sizeof(cd) * tot, hipMemcpyHostToDevice));
        HIPTRY(hipMemsetAsync(pl->d_status, 0, sizeof(int) * B, st));
        HIPTRY(hipMemsetAsync(pl->d_iwork, 0, sizeof(int) * (4 * B + KB_QUEUE_WORDS), st));     // member-queue counters
        HIPTRY(hipMemsetAsync(pl->d_team, 0, sizeof(TeamCtl) * B, st));
        HIPTRY(hipMemsetAsync(pl->d_needqr, 0, sizeof(int) * B, st));
        Chunk& ch = pl->chunks[0];
        hipLaunchKernelGGL(k_transpose_in, dim3(64, B), dim3(256), 0, st, pl->d_items, d_dense, pl->d_arena, KB_BUF_P, 1);
        hipLaunchKernelGGL(k_fill_ones, dim3(B), dim3(256), 0, st, pl->d_items, pl->d_varena);
        if ((r = launch_eig(pl, ch, nullptr))) break;
        launch_gemm<3>(pl, ch, ch.lmax, ch.lmax);   // P = Qh X (Dsqi = 1)
        hipLaunchKernelGGL(k_transpose_out, dim3(64, B), dim3(256), 0, st, pl->d_items, pl->d_arena, KB_BUF_P, d_dense, 1);
        HIPTRY(hipStreamSynchronize(st));
        if (P) HIPTRY(hipMemcpy(P, d_dense, sizeof(cd) * tot, hipMemcpyDeviceToHost));
        if (mu) HIPTRY(hipMemcpy(mu, pl->d_mu, sizeof(cd) * pl->total_lines, hipMemcpyDeviceToHost));
        if (status) HIPTRY(hipMemcpy(status, pl->d_status, sizeof(int) * B, hipMemcpyDeviceToHost));
        ctx->last_eig_fallbacks = kbdm_plan_eig_fallbacks(pl);
        HIPTRY(hipGetLastError());
    } while (0);
    hipFree(d_dense);
    kbdm_plan_destroy(pl);
    return r;
}

int kbdm_ctx_last_eig_fallbacks(kbdm_ctx* ctx) { return ctx ? ctx->last_eig_fallbacks : -1; }

// ---------------------------------------------------------------- rows next to the hot path
int kbdm_rmse_batch(kbdm_ctx* ctx, const double* data, int N, double dwell, const double* lines,
                    const int64_t* cand_off, int ncand, double* rmse_out) {
    if (!ctx || !data || !cand_off || !rmse_out || N < 1 || ncand < 0) return fail(KBDM_E_INVALID, "bad rmse arguments");
    if (ncand == 0) return KBDM_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t nrows = cand_off[ncand];
    if (nrows < 0 || (nrows > 0 && !lines)) return fail(KBDM_E_INVALID, "bad candidate offsets");
    for (int c = 0; c < ncand; ++c)
        if (cand_off[c + 1] < cand_off[c]) return fail(KBDM_E_INVALID, "candidate offsets must be non-decreasing");
    cd *d_data = nullptr, *d_res = nullptr;
    double *d_lines = nullptr, *d_out = nullptr;
    long long* d_off = nullptr;
    hipStream_t st = ctx->stream;
    int r = KBDM_OK;
    do {
        if (hipMalloc(&d_data, sizeof(cd) * N) != hipSuccess || hipMalloc(&d_res, sizeof(cd) * (size_t)N * ncand) != hipSuccess ||
            hipMalloc(&d_lines, sizeof(double) * 4 * std::max<int64_t>(nrows, 1)) != hipSuccess ||
            hipMalloc(&d_off, sizeof(long long) * (ncand + 1)) != hipSuccess || hipMalloc(&d_out, sizeof(double) * ncand) != hipSuccess) {
            r = fail(KBDM_E_NOMEM, "hipMalloc (rmse)");
            break;
        }
        hipMemcpyAsync(d_data, data, sizeof(cd) * N, hipMemcpyHostToDevice, st);
        if (nrows > 0) hipMemcpyAsync(d_lines, lines, sizeof(double) * 4 * nrows, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_off, cand_off, sizeof(long long) * (ncand + 1), hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(k_rmse, dim3(ncand), dim3(256), 0, st, d_data, N, dwell, d_lines, d_off, d_res, d_out);
        hipMemcpyAsync(rmse_out, d_out, sizeof(double) * ncand, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { r = fail(KBDM_E_HIP, "rmse kernel failed"); break; }
        for (int c = 0; c < ncand; ++c)
            if (cand_off[c + 1] == cand_off[c]) rmse_out[c] = std::numeric_limits<double>::infinity();
    } while (0);
    hipFree(d_data); hipFree(d_res); hipFree(d_lines); hipFree(d_off); hipFree(d_out);
    return r;
}

int kbdm_silhouette_samples(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* labels, double* out) {
    if (!ctx || !X || !labels || !out || n < 2 || dim < 1 || dim > KB_SIL_MAXDIM) return fail(KBDM_E_INVALID, "bad silhouette arguments");
    HIPCHK(hipSetDevice(ctx->device));
    // classes = sorted distinct labels; samples sorted by class (stable)
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return labels[a] < labels[b]; });
    std::vector<int> cls(n), cstart;
    int nclass = 0;
    for (int k = 0; k < n; ++k) {
        if (k == 0 || labels[order[k]] != labels[order[k - 1]]) { cstart.push_back(k); ++nclass; }
        cls[k] = nclass - 1;
    }
    cstart.push_back(n);
    if (nclass < 2 || nclass > n - 1) return fail(KBDM_E_INVALID, "number of labels must be in 2 .. n_samples - 1");
    std::vector<double> xs((size_t)n * dim), so(n);
    for (int k = 0; k < n; ++k) memcpy(&xs[(size_t)k * dim], &X[(size_t)order[k] * dim], sizeof(double) * dim);
    double *d_x = nullptr, *d_o = nullptr;
    int *d_cls = nullptr, *d_cs = nullptr;
    hipStream_t st = ctx->stream;
    int r = KBDM_OK;
    do {
        if (hipMalloc(&d_x, sizeof(double) * n * dim) != hipSuccess || hipMalloc(&d_o, sizeof(double) * n) != hipSuccess ||
            hipMalloc(&d_cls, sizeof(int) * n) != hipSuccess || hipMalloc(&d_cs, sizeof(int) * (nclass + 1)) != hipSuccess) {
            r = fail(KBDM_E_NOMEM, "hipMalloc (silhouette)");
            break;
        }
        hipMemcpyAsync(d_x, xs.data(), sizeof(double) * n * dim, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_cls, cls.data(), sizeof(int) * n, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_cs, cstart.data(), sizeof(int) * (nclass + 1), hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(k_silhouette, dim3((n + KB_SIL_TILE - 1) / KB_SIL_TILE), dim3(KB_SIL_TILE), 0, st, d_x, n, dim,
                           d_cls, d_cs, nclass, d_o);
        hipMemcpyAsync(so.data(), d_o, sizeof(double) * n, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { r = fail(KBDM_E_HIP, "silhouette kernel failed"); break; }
        for (int k = 0; k < n; ++k) out[order[k]] = so[k];
    } while (0);
    hipFree(d_x); hipFree(d_o); hipFree(d_cls); hipFree(d_cs);
    return r;
}

int kbdm_silhouette_sweep(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* labels, int nfits, double* out,
                          int32_t* valid_out) {
    if (!ctx || !X || !labels || !out || n < 2 || dim < 1 || dim > KB_SIL_MAXDIM || nfits < 1) return fail(KBDM_E_INVALID, "bad silhouette arguments");
    HIPCHK(hipSetDevice(ctx->device));
    // per labeling: classes = sorted distinct labels, samples sorted by class (stable) - as kbdm_silhouette_samples does
    std::vector<int> order((size_t)nfits * n), cls((size_t)nfits * n), cstart, coff(nfits), nclass(nfits);
    const int nthreads = std::max(1, std::min(nfits, std::min(16, (int)std::thread::hardware_concurrency())));
    std::vector<std::vector<int>> cs(nfits);
    {
        std::vector<std::thread> pool;
        for (int tix = 0; tix < nthreads; ++tix)
            pool.emplace_back([&, tix]() {
                for (int f = tix; f < nfits; f += nthreads) {
                    const int32_t* lab = labels + (size_t)f * n;
                    int* ord = order.data() + (size_t)f * n;
                    int* cl = cls.data() + (size_t)f * n;
                    std::iota(ord, ord + n, 0);
                    std::stable_sort(ord, ord + n, [&](int a, int b) { return lab[a] < lab[b]; });
                    int nc = 0;
                    for (int k = 0; k < n; ++k) {
                        if (k == 0 || lab[ord[k]] != lab[ord[k - 1]]) { cs[f].push_back(k); ++nc; }
                        cl[k] = nc - 1;
                    }
                    cs[f].push_back(n);
                    nclass[f] = (nc < 2 || nc > n - 1) ? 0 : nc;      // sklearn's precondition: such a labeling has no silhouettes
                }
            });
        for (auto& th : pool) th.join();
    }
    for (int f = 0; f < nfits; ++f) {
        coff[f] = (int)cstart.size();
        cstart.insert(cstart.end(), cs[f].begin(), cs[f].end());
        if (valid_out) valid_out[f] = nclass[f] ? 1 : 0;
    }
    double *d_x = nullptr, *d_o = nullptr;
    int *d_ord = nullptr, *d_cls = nullptr, *d_cs = nullptr, *d_coff = nullptr, *d_nc = nullptr;
    hipStream_t st = ctx->stream;
    int r = KBDM_OK;
    do {
        if (hipMalloc(&d_x, sizeof(double) * n * dim) != hipSuccess || hipMalloc(&d_o, sizeof(double) * (size_t)nfits * n) != hipSuccess ||
            hipMalloc(&d_ord, sizeof(int) * (size_t)nfits * n) != hipSuccess || hipMalloc(&d_cls, sizeof(int) * (size_t)nfits * n) != hipSuccess ||
            hipMalloc(&d_cs, sizeof(int) * cstart.size()) != hipSuccess || hipMalloc(&d_coff, sizeof(int) * nfits) != hipSuccess ||
            hipMalloc(&d_nc, sizeof(int) * nfits) != hipSuccess) {
            r = fail(KBDM_E_NOMEM, "hipMalloc (silhouette sweep)");
            break;
        }
        hipMemcpyAsync(d_x, X, sizeof(double) * n * dim, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_ord, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_cls, cls.data(), sizeof(int) * cls.size(), hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_cs, cstart.data(), sizeof(int) * cstart.size(), hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_coff, coff.data(), sizeof(int) * nfits, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_nc, nclass.data(), sizeof(int) * nfits, hipMemcpyHostToDevice, st);
        hipMemsetAsync(d_o, 0, sizeof(double) * (size_t)nfits * n, st);
        hipLaunchKernelGGL(k_silhouette_sweep, dim3((n + KB_SIL_TILE - 1) / KB_SIL_TILE, nfits), dim3(KB_SIL_TILE), 0, st, d_x, n, dim,
                           d_ord, d_cls, d_cs, d_coff, d_nc, d_o);
        hipMemcpyAsync(out, d_o, sizeof(double) * (size_t)nfits * n, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { r = fail(KBDM_E_HIP, "silhouette kernel failed"); break; }
    } while (0);
    hipFree(d_x); hipFree(d_o); hipFree(d_ord); hipFree(d_cls); hipFree(d_cs); hipFree(d_coff); hipFree(d_nc);
    return r;
}

int kbdm_hdbscan_labels_from_mst(int n, const int32_t* a, const int32_t* b, const double* w, int min_cluster_size,
                                 int32_t* labels_out) {
    if (n < 1 || !labels_out || (n > 1 && (!a || !b || !w)) || min_cluster_size < 2) return fail(KBDM_E_INVALID, "bad mst arguments");
    std::vector<MstEdge> e(std::max(n - 1, 0));
    for (int i = 0; i < n - 1; ++i) {
        if (a[i] < 0 || a[i] >= n || b[i] < 0 || b[i] >= n) return fail(KBDM_E_INVALID, "mst edge out of range");
        e[i] = MstEdge{a[i], b[i], w[i]};
    }
    return hdbscan_labels_from_mst(n, e.data(), min_cluster_size, labels_out);
}

// The k-nearest-neighbour pass on stream st: knn[i * K + q], q < K, ascending (kbdm_next.hpp).  d_lo / d_skip: n entries each,
// only needed (and only touched) when the lists do not fit one pass; knn_passes() tells.
namespace {
struct KnnGeom { int tpb, Kp, npass; };
KnnGeom knn_geom(int K) {
    // threads (= samples) per workgroup: as many as keep the running lists in LDS; lists longer than 8 threads' worth of
    // LDS are produced in passes of Kp entries
    KnnGeom g{KB_KNN_TPB, K, 1};
    auto bytes = [&](int k, int t) { return ((size_t)k * t + (size_t)t * KB_SIL_MAXDIM) * sizeof(double); };
    while (g.tpb > 8 && bytes(K, g.tpb) > (size_t)LDS_MAX - 64) g.tpb >>= 1;
    if (bytes(K, g.tpb) > (size_t)LDS_MAX - 64)
        g.Kp = (int)((((size_t)LDS_MAX - 64) / sizeof(double) - (size_t)g.tpb * KB_SIL_MAXDIM) / g.tpb);
    g.npass = (K + g.Kp - 1) / g.Kp;
    return g;
}
void launch_knn(const KnnGeom& g, hipStream_t st, const double* d_x, int n, int dim, int K, double* d_knn, double* d_lo, int* d_skip) {
    static const int use_select = env_int("KBDM_KNN_SELECT", 1);
    // selection with a workgroup per sample (kbdm_next.hpp); a handful of neighbours is cheaper by insertion (5.6 ms against 13 at
    // K = 1 and 20 000 samples, where K = 150 costs the insertion 85 ms and the selection 11)
    if (use_select && K >= 16 && K <= KB_KNN_SEL_MAXK && K <= n) {
        int kp2 = 1;
        while (kp2 < K) kp2 <<= 1;
        hipLaunchKernelGGL(k_knn_select, dim3(n), dim3(256), sizeof(double) * kp2, st, d_x, n, dim, K, d_knn);
        return;
    }
    for (int koff = 0; koff < K; koff += g.Kp) {
        const int kp = std::min(g.Kp, K - koff);
        const size_t lds = ((size_t)kp * g.tpb + (size_t)g.tpb * KB_SIL_MAXDIM) * sizeof(double);
        hipLaunchKernelGGL(k_knn_dist, dim3((n + g.tpb - 1) / g.tpb), dim3(g.tpb), lds, st, d_x, n, dim, K, d_knn, koff, kp, d_lo, d_skip);
    }
}
}  // namespace

int kbdm_core_distances(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* min_samples, int nfits, double* out) {
    if (!ctx || !X || !min_samples || !out || n < 2 || dim < 1 || dim > KB_SIL_MAXDIM || nfits < 1)
        return fail(KBDM_E_INVALID, "bad core-distance arguments");
    int K = 1;
    for (int f = 0; f < nfits; ++f) {
        if (min_samples[f] < 1 || min_samples[f] > n) return fail(KBDM_E_INVALID, "min_samples out of range");
        K = std::max(K, (int)min_samples[f]);
    }
    const KnnGeom g = knn_geom(K);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_dist), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    double *d_x = nullptr, *d_knn = nullptr, *d_lo = nullptr;
    int* d_skip = nullptr;
    hipStream_t st = ctx->stream;
    std::vector<double> knn((size_t)n * K);
    int r = KBDM_OK;
    do {
        if (hipMalloc(&d_x, sizeof(double) * n * dim) != hipSuccess || hipMalloc(&d_knn, sizeof(double) * (size_t)n * K) != hipSuccess ||
            (g.npass > 1 && (hipMalloc(&d_lo, sizeof(double) * n) != hipSuccess || hipMalloc(&d_skip, sizeof(int) * n) != hipSuccess))) {
            r = fail(KBDM_E_NOMEM, "hipMalloc (core distances)");
            break;
        }
        hipMemcpyAsync(d_x, X, sizeof(double) * n * dim, hipMemcpyHostToDevice, st);
        launch_knn(g, st, d_x, n, dim, K, d_knn, d_lo, d_skip);
        hipMemcpyAsync(knn.data(), d_knn, sizeof(double) * knn.size(), hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { r = fail(KBDM_E_HIP, "k-nearest-neighbour kernel failed"); break; }
    } while (0);
    hipFree(d_x); hipFree(d_knn); hipFree(d_lo); hipFree(d_skip);
    if (r) return r;
    for (int f = 0; f < nfits; ++f)
        for (int i = 0; i < n; ++i) out[(size_t)f * n + i] = knn[(size_t)i * K + min_samples[f] - 1];
    return KBDM_OK;
}

int kbdm_hdbscan_sweep(kbdm_ctx* ctx, const double* X, int n, int dim, const int32_t* min_samples, int nfits,
                       int min_cluster_size, int32_t* labels_out, int32_t* nclusters_out) {
    if (!ctx || !X || !min_samples || !labels_out || n < 2 || dim < 1 || dim > KB_SIL_MAXDIM || nfits < 1 || min_cluster_size < 2)
        return fail(KBDM_E_INVALID, "bad hdbscan arguments");
    int K = 1;
    for (int f = 0; f < nfits; ++f) {
        if (min_samples[f] < 1 || min_samples[f] > n) return fail(KBDM_E_INVALID, "min_samples out of range");
        K = std::max(K, (int)min_samples[f]);
    }
    const KnnGeom g = knn_geom(K);
    const int npass = g.npass;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_dist), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 64));
    double *d_x = nullptr, *d_knn = nullptr, *d_best = nullptr, *d_core = nullptr, *d_lo = nullptr;
    int *d_ms = nullptr, *d_src = nullptr, *d_skip = nullptr;
    KbEdge* d_edges = nullptr;
    hipStream_t st = ctx->stream;
    std::vector<KbEdge> edges((size_t)nfits * (n - 1));
    int r = KBDM_OK;
    do {
        if (hipMalloc(&d_x, sizeof(double) * n * dim) != hipSuccess || hipMalloc(&d_knn, sizeof(double) * (size_t)n * K) != hipSuccess ||
            hipMalloc(&d_best, sizeof(double) * (size_t)n * nfits) != hipSuccess || hipMalloc(&d_core, sizeof(double) * (size_t)n * nfits) != hipSuccess ||
            hipMalloc(&d_src, sizeof(int) * (size_t)n * nfits) != hipSuccess ||
            hipMalloc(&d_ms, sizeof(int) * nfits) != hipSuccess || hipMalloc(&d_edges, sizeof(KbEdge) * (size_t)nfits * (n - 1)) != hipSuccess ||
            (npass > 1 && (hipMalloc(&d_lo, sizeof(double) * n) != hipSuccess || hipMalloc(&d_skip, sizeof(int) * n) != hipSuccess))) {
            r = fail(KBDM_E_NOMEM, "hipMalloc (hdbscan)");
            break;
        }
        hipMemcpyAsync(d_x, X, sizeof(double) * n * dim, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_ms, min_samples, sizeof(int) * nfits, hipMemcpyHostToDevice, st);
        launch_knn(g, st, d_x, n, dim, K, d_knn, d_lo, d_skip);
        // Prim per fit: the register-resident form for the reference's 4-dimensional samples while a thread's share fits
        // its registers (n <= 20480: a C2 ensemble pools about 20 000 lines), the general form beyond
        const size_t psm = sizeof(int) * (size_t)n;
        auto prim_reg = [&](auto kern) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX - 1024);
            hipLaunchKernelGGL(kern, dim3(nfits), dim3(1024), psm, st, d_x, n, K, d_knn, d_ms, d_core, d_edges);
        };
        static const int exp_stream = env_int("KBDM_PRIM_STREAM", 0);
        if (exp_stream && dim == 4 && n <= 1024 * 20) prim_reg(k_prim_mst_reg<20, false>);
        else if (dim == 4 && n <= 1024 * 6) prim_reg(k_prim_mst_reg<6>);
        else if (dim == 4 && n <= 1024 * 12) prim_reg(k_prim_mst_reg<12>);
        else if (dim == 4 && n <= 1024 * 20) prim_reg(k_prim_mst_reg<20>);
        else if (dim == 4 && n <= 1024 * 30) prim_reg(k_prim_mst_reg<30, false>);
        else if (dim == 4 && n <= 40000) prim_reg(k_prim_mst_reg<40, false>);
        else {
            hipLaunchKernelGGL(k_prim_mst, dim3(nfits), dim3(1024), 0, st, d_x, n, dim, K, d_knn, d_ms, d_best, d_core, d_src, d_edges);
        }
        hipMemcpyAsync(edges.data(), d_edges, sizeof(KbEdge) * edges.size(), hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { r = fail(KBDM_E_HIP, "hdbscan kernels failed"); break; }
    } while (0);
    hipFree(d_x); hipFree(d_knn); hipFree(d_best); hipFree(d_core); hipFree(d_src); hipFree(d_ms); hipFree(d_edges); hipFree(d_lo); hipFree(d_skip);
    if (r) return r;
    // the trees: independent per fit, a few host threads
    const int nthreads = std::max(1, std::min(nfits, std::min(16, (int)std::thread::hardware_concurrency())));
    std::vector<std::thread> pool;
    for (int tix = 0; tix < nthreads; ++tix)
        pool.emplace_back([&, tix]() {
            std::vector<MstEdge> e(n - 1);
            for (int f = tix; f < nfits; f += nthreads) {
                const KbEdge* src = edges.data() + (size_t)f * (n - 1);
                for (int i = 0; i < n - 1; ++i) e[i] = MstEdge{src[i].a, src[i].b, src[i].w};
                const int nc = hdbscan_labels_from_mst(n, e.data(), min_cluster_size, labels_out + (size_t)f * n);
                if (nclusters_out) nclusters_out[f] = nc;
            }
        });
    for (auto& th : pool) th.join();
    return KBDM_OK;
}

}  // extern "C"
