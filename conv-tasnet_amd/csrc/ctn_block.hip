// Composite entry points: a whole stack of gLN TemporalBlocks behind ONE call of the C ABI.
//
// The reference runs `temporal_conv_net = nn.Sequential(*repeats)` (src/conv_tasnet.py:176-186): X*R TemporalBlocks
// (:218-278), each  x + pw2(gLN(prelu(dw(gLN(prelu(pw1(x)))))))  -- one Python call per module.  Here the host issues
// every launch of the stack (3 per block forward, 6 + 2 weight-gradient launches per block backward) from C++, so the
// Python side of a training step makes ~50 calls instead of ~450 and the GPU queues never run dry (round 1: the host
// needed 12.7 ms to enqueue a 15.8 ms step).  The per-kernel entry points stay the unit-tested surface; these
// composites call exactly those, in the order ops.GlnBlock does, so results are bitwise identical.
//
// Streams: the two weight-gradient GEMMs of a block feed only the optimiser; with side_stream != NULL they are issued
// there behind a device-scope event (ctn_stream_order) and overlap the HBM-bound kernels of the backward chain.  The
// call ends by ordering `stream` after `side_stream`, so every gradient is complete in stream order when it returns.
#include "ctn_common.h"
#include "../../include/ctn_hip.h"
#include <mutex>
#include <stdlib.h>
#include <vector>


// ---- measurement hook (bench.py's roofline leg): HIP events around every launch group of the composite stacks, on the stream
// the group is launched to.  Off by default (no events, no overhead); ctn_probe_enable(1) starts a recording, ctn_probe_read
// waits for the recorded events, returns (family id, microseconds) per launch group in issue order and ends the recording.
enum { F_K1 = 0, F_K2, F_K3, F_B1, F_B2, F_B3, F_B4, F_B5, F_B6, F_FIN, F_PREP, F_CLN_FWD, F_CLN_BWD, F_TAPS, F_WFLUSH, F_FRAME, F_COUNT };
namespace {
struct ProbeRec { int fam; hipEvent_t e0, e1; };
std::vector<ProbeRec> g_probe;          // guarded by g_probe_mu: a second host thread that drives the library while a recording is
std::mutex g_probe_mu;                  // on appends its launch groups too (they are attributed by family, not by thread)
bool g_probe_on = false;
struct ProbeMark {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st;
    int fam;
    ProbeMark(int f, void* stream) : st((hipStream_t)stream), fam(f) {
        if (!g_probe_on) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
        hipEventRecord(e0, st);
    }
    ~ProbeMark() {
        if (e0 == nullptr) return;
        hipEventRecord(e1, st);
        std::lock_guard<std::mutex> lk(g_probe_mu);
        g_probe.push_back(ProbeRec{fam, e0, e1});
    }
};
}  // namespace
#define PROBED(fam, stream, expr) [&]() -> int { ProbeMark pm_(fam, stream); return (expr); }()

extern "C" int ctn_probe_enable(int on) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    for (ProbeRec& r : g_probe) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    g_probe.clear();
    g_probe_on = on != 0;
    return CTN_OK;
}

extern "C" int ctn_probe_read(int* fam, float* us, int cap) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    int n = 0;
    for (ProbeRec& r : g_probe) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1.f;
        if (n < cap && fam && us) { fam[n] = r.fam; us[n] = ms * 1e3f; }
        ++n;
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    g_probe.clear();
    g_probe_on = false;
    return n;
}

// lab builds only (-DCTN_EXP_SKIP; never in the product library): ctn_tune("exp_skip", mask) for TIMING experiments whose results are
// wrong -- bit 0 = the gLN stack's backward skips B4 (profiles/README.md r04_j)
#ifdef CTN_EXP_SKIP
int g_ctn_exp_skip = 0;
#else
static constexpr int g_ctn_exp_skip = 0;
#endif
// ctn_tune("bwd_events", 1 | 2): cross-stream events per block of the backward pass.  2 (rounds 1-4): the second 1x1 conv's weight
// gradient is forked behind B1, the first one's behind B4.  1: ONE fork per block, behind B5 -- dW2 of block i needs only dy_i (the
// output of B5 of block i+1) and forward tensors, so it rides behind the event that releases dW1 of block i+1; each event costs the
// main chain ~5 us (profiles/README.md r04_k).  0 (default) = the measured best per stack: gLN 1 (paper step 10.19 -> 10.14 ms), cLN 2
// (11.44 vs 11.79 ms with one fork: its second stream carries the latency-bound sums that the early dW2 then queues behind).
// CTN_BWD_EVENTS=1|2 at first use overrides it (fresh-process A/B).
int g_ctn_bwd_events = -1;
static int bwd_events(int dflt) {
    if (g_ctn_bwd_events < 0) {
        const char* e = getenv("CTN_BWD_EVENTS");
        g_ctn_bwd_events = (e && (*e == '1' || *e == '2')) ? *e - '0' : 0;
    }
    return g_ctn_bwd_events ? g_ctn_bwd_events : dflt;
}

namespace {

enum { P_W1 = 0, P_A1, P_G1, P_B1, P_D, P_A2, P_G2, P_B2, P_W2, NPARAM };

inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

// One slot of the per-block weight region: an fp32 [I, O] copy (fp32 arithmetic) or the bf16 piece fragments of the same
// operand (b3 arithmetic, ctn_split_b3_batch) -- the same bytes when H and B are multiples of 32.
inline size_t wslot_bytes(int B, int H) {
    size_t n = (size_t)H * B * sizeof(float);
    const size_t a = ctn_split_b3_bytes(H, B), b = ctn_split_b3_bytes(B, H), c = ctn_split_h3_bytes(H, B), d = ctn_split_h3_bytes(B, H);
    if (a > n) n = a;
    if (b > n) n = b;
    if (c > n) n = c;
    if (d > n) n = d;
    return align256(n);
}
inline bool pieces(int R) { return ctn_gemm_arith() != 0 && R >= 64; }      // the rule of ctn_pw_gemm(trans_w = 2)
// h3 arithmetic (include/ctn_hip.h): the six GEMMs of every block on the ctn_*_h3 entry points, operand ranges tracked by the
// producing kernels.  Both layer widths must qualify; otherwise the stack runs as b6.
inline bool use_h3(int B, int H) { return ctn_gemm_arith() == 3 && B >= 64 && H >= 64; }

// Weight operands of every block, prepared once per call into region [nblocks][2 slots]: slot 0 for the GEMM with H output
// rows, slot 1 for the one with B output rows.  fwd: (w1 -> H rows, w2 -> B rows) as stored; bwd: (w2 -> H rows, w1 -> B rows)
// transposed.  tw_h / tw_b receive the trans_w code of ctn_pw_gemm for the prepared operand (2 pieces, 1 fp32 [I, O] copy;
// backward without pieces uses the stored matrices: no copy, code 1).
// h3: pieces from ctn_split_h3_batch (code 3), and gb [nblocks][2] receives {max |gamma2|, max |beta2|} of every block.
int prepare_weights(const void* const* params, int nblocks, int B, int H, bool backward, char* region, size_t slot,
                    int* tw_h, int* tw_b, bool h3, float* gb, void* stream) {
    std::vector<const void*> src(nblocks);
    std::vector<void*> dst(nblocks);
    int rc;
    for (int half = 0; half < 2; ++half) {              // 0: the H-row GEMM, 1: the B-row GEMM
        const int R = half == 0 ? H : B, Cn = half == 0 ? B : H;
        const int pidx = (half == 0) != backward ? P_W1 : P_W2;      // fwd: H rows <- w1, B rows <- w2; bwd: H rows <- w2, B rows <- w1
        for (int i = 0; i < nblocks; ++i) {
            src[i] = ((const void* const*)(params + (size_t)i * NPARAM))[pidx];
            dst[i] = region + ((size_t)2 * i + half) * slot;
        }
        int* const tw = half == 0 ? tw_h : tw_b;
        if (h3) {
            if ((rc = PROBED(F_PREP, stream, ctn_split_h3_batch(src.data(), dst.data(), nblocks, R, Cn, backward ? 1 : 0, stream)))) return rc;
            *tw = 3;
        } else if (pieces(R)) {
            if ((rc = PROBED(F_PREP, stream, ctn_split_b3_batch(src.data(), dst.data(), nblocks, R, Cn, backward ? 1 : 0, stream)))) return rc;
            *tw = 2;
        } else if (!backward) {
            if ((rc = PROBED(F_PREP, stream, ctn_transpose_batch(src.data(), dst.data(), nblocks, R, Cn, stream)))) return rc;     // [R, Cn] -> [Cn, R]
            *tw = 1;
        } else {
            *tw = -1;       // use the stored matrix (trans_w = 1)
        }
    }
    if (h3 && gb != nullptr) {      // (the cLN stacks have no operand prologue)
        std::vector<const void*> gsrc(2 * (size_t)nblocks);
        std::vector<void*> gdst(2 * (size_t)nblocks);
        for (int i = 0; i < nblocks; ++i) {
            gsrc[2 * i] = ((const void* const*)(params + (size_t)i * NPARAM))[P_G2];
            gsrc[2 * i + 1] = ((const void* const*)(params + (size_t)i * NPARAM))[P_B2];
            gdst[2 * i] = gb + 2 * i;
            gdst[2 * i + 1] = gb + 2 * i + 1;
        }
        if ((rc = PROBED(F_PREP, stream, ctn_absmax_batch(gsrc.data(), gdst.data(), 2 * nblocks, H, stream)))) return rc;
    }
    return CTN_OK;
}

struct FwdWs {
    size_t st1, st2, wt, gb, total;
    int np1;
};
FwdWs fwd_ws(int M, int B, int H, int Kp, int nblocks) {
    FwdWs w;
    w.np1 = ctn_pw_stats_parts(M, H, Kp);
    w.st1 = 0;
    w.st2 = align256((size_t)M * w.np1 * 2 * sizeof(double));
    w.wt = w.st2 + align256((size_t)M * H * 2 * sizeof(double));
    w.gb = w.wt + (size_t)nblocks * 2 * wslot_bytes(B, H);                               // [nblocks][w1 operand | w2 operand]
    w.total = w.gb + align256((size_t)nblocks * 2 * sizeof(float));                      // h3: {max |gamma2|, max |beta2|} per block
    return w;
}

struct BwdWs {
    size_t dn2, s2p, s1p, pc, da1p, slab, wp, gb, amax, total, pc_slot, da1p_slot, s1p_slot;
    size_t slab_bytes;
    int np2;
};
BwdWs bwd_ws(int M, int B, int H, int Kp, int P, int nblocks) {
    BwdWs w;
    w.np2 = ctn_pw_stats_parts(M, H, Kp);
    size_t o = 0;
    w.dn2 = o; o += align256((size_t)M * H * Kp * sizeof(float));
    w.s2p = o; o += align256((size_t)M * w.np2 * 8 * sizeof(double));        // (8 sums per tile with ctn_tune("gln_fuse", 1), else 2)
    w.s1p_slot = align256((size_t)M * H * 2 * sizeof(double));        // per block: the fused weight gradient of the second stream reads it
    w.s1p = o; o += (size_t)nblocks * w.s1p_slot;
    // per-block slots: the finalize kernel of block i runs on the weight-gradient stream while the chain is already in block i-1
    w.pc_slot = align256((size_t)ctn_dw_bwd_rows(P, 3) * M * H * sizeof(float));      // (a row more than fused = 1: the dalpha1 partials of gln_fuse)
    w.da1p_slot = align256((size_t)M * H * sizeof(float));          // PReLU-slope partials of ctn_gln_prelu_bwd: one per (m, channel)
    w.pc = o; o += (size_t)nblocks * w.pc_slot;
    w.da1p = o; o += (size_t)nblocks * w.da1p_slot;
    const size_t s1 = ctn_pw_wgrad_workspace(M, H, B, Kp), s2 = ctn_pw_wgrad_workspace(M, B, H, Kp);
    w.slab_bytes = s1 > s2 ? s1 : s2;
    w.slab = o; o += 2 * align256(w.slab_bytes);          // two: a chained weight gradient writes one while the next launch sums the other
    w.wp = o; o += (size_t)nblocks * 2 * wslot_bytes(B, H);         // [nblocks][w2 operand (H rows) | w1 operand (B rows)], b3 pieces
    w.gb = o; o += align256((size_t)nblocks * 2 * sizeof(float));   // h3: {max |gamma2|, max |beta2|} per block
    w.amax = o; o += align256((size_t)nblocks * 2 * M * CTN_AMAX_SLOTS * sizeof(unsigned));     // h3: tracked maxima of (dy, dh1) per block and utterance
    w.total = o;
    return w;
}

}  // namespace

extern "C" {

size_t ctn_tcn_gln_fwd_workspace(int M, int B, int H, int Kp, int nblocks) { return fwd_ws(M, B, H, Kp, nblocks).total; }
size_t ctn_tcn_gln_bwd_workspace(int M, int B, int H, int Kp, int P, int nblocks) { return bwd_ws(M, B, H, Kp, P, nblocks).total; }

int ctn_tcn_gln_fwd(const void* const* params, const int* dilation, int nblocks, const float* x0,
                    float* xs, float* h1s, float* ds, float* ms, unsigned* amax, int save,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream) {
    CTN_REQUIRE(params && dilation && nblocks > 0 && x0 && xs && h1s && ds && ms && workspace, "ctn_tcn_gln_fwd: null pointer");
    const bool h3 = use_h3(B, H);
    CTN_REQUIRE(!h3 || amax, "ctn_tcn_gln_fwd: the h3 arithmetic needs the amax array");
    CTN_REQUIRE(M > 0 && B > 0 && H > 0 && K > 0 && Kp >= K && P >= 1, "ctn_tcn_gln_fwd: bad sizes");
    const FwdWs w = fwd_ws(M, B, H, Kp, nblocks);
    if (workspace_bytes < w.total) {
        ctn_set_error("ctn_tcn_gln_fwd: workspace too small (%zu < %zu)", workspace_bytes, w.total);
        return CTN_ERR_WORKSPACE;
    }
    double* const st1 = (double*)((char*)workspace + w.st1);
    double* const st2 = (double*)((char*)workspace + w.st2);
    const size_t xsz = (size_t)M * B * Kp, hsz = (size_t)M * H * Kp;
    // the weight operand of both 1x1 convolutions of every block, prepared once: bf16 piece fragments (split-bf16 arithmetics) or
    // [I, O] fp32 copies (16-byte LDS row writes instead of the transposing scatter); two launches for the whole stack
    char* const wreg = (char*)workspace + w.wt;
    const size_t slot = wslot_bytes(B, H);
    int tw1 = 0, tw2 = 0, rc;
    for (int i = 0; i < nblocks; ++i) {
        const float* const* p = (const float* const*)(params + (size_t)i * NPARAM);
        for (int j = 0; j < NPARAM; ++j) CTN_REQUIRE(p[j], "ctn_tcn_gln_fwd: block %d parameter %d is null", i, j);
    }
    float* const gb = (float*)((char*)workspace + w.gb);
    if ((rc = prepare_weights(params, nblocks, B, H, false, wreg, slot, &tw1, &tw2, h3, gb, stream))) return rc;
    if (h3) {       // range tracking: every slot starts at 0; the stack's input is measured here, everything else by its producer
        if (hipMemsetAsync(amax, 0, (size_t)nblocks * 2 * M * CTN_AMAX_SLOTS * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) {
            ctn_set_error("ctn_tcn_gln_fwd: hipMemsetAsync failed");
            return CTN_ERR_LAUNCH;
        }
        if ((rc = PROBED(F_PREP, stream, ctn_absmax_rows(x0, M, (long long)B * Kp, amax, stream)))) return rc;
    }
    // Two chains: the forward pass of a block is three strictly dependent kernels (K1 -> statistics -> K2 -> statistics -> K3) with
    // nothing beside them, so the fill and epilogue of every GEMM and the whole HBM-bound depthwise kernel run with the matrix
    // cores idle.  Utterances are independent (gLN statistics are per utterance): with side_stream the batch is cut in two halves
    // that run the same chain on two streams, so that one half's HBM-bound phases sit beside the other half's MFMA phases.  Every
    // kernel computes utterance by utterance, so the values are bitwise those of the single chain.
    const int nch = (side_stream != nullptr && M >= 2) ? 2 : 1;
    const int m0c[2] = {0, M / 2}, mcc[2] = {nch == 2 ? M / 2 : M, M - M / 2};
    void* const sts[2] = {stream, side_stream};
    if (nch == 2 && (rc = ctn_stream_order(stream, side_stream))) return rc;        // x0 and the weight operands are ready
    for (int i = 0; i < nblocks; ++i) {
        const float* const* p = (const float* const*)(params + (size_t)i * NPARAM);
        const float* const w1t = (const float*)(wreg + (size_t)(2 * i) * slot);
        const float* const w2t = (const float*)(wreg + (size_t)(2 * i + 1) * slot);
        // save = 0 (inference): one h1 / d slot and two ping-pong x slots; save = 1: a slot per block for the backward pass
        const float* const xin = i == 0 ? x0 : xs + (save ? (size_t)(i - 1) : (size_t)((i - 1) & 1)) * xsz;
        float* const h1 = h1s + (save ? (size_t)i * hsz : 0);
        float* const d = ds + (save ? (size_t)i * hsz : 0);
        float* const out = xs + (save ? (size_t)i : (size_t)(i & 1)) * xsz;
        float* const ms1 = ms + ((size_t)(save ? i : 0) * 2 + 0) * M * 2;
        float* const ms2 = ms + ((size_t)(save ? i : 0) * 2 + 1) * M * 2;
        for (int step = 0; step < 3; ++step)             // issue K1 of both chains, then K2 of both, then K3 of both: both queues stay fed
            for (int c = 0; c < nch; ++c) {
                const int m0 = m0c[c], Mc = mcc[c];
                void* const st = sts[c];
                const size_t xo = (size_t)m0 * B * Kp, ho = (size_t)m0 * H * Kp;
                double* const s1 = st1 + (size_t)m0 * w.np1 * 2;
                double* const s2 = st2 + (size_t)m0 * H * 2;
                unsigned* const ax = h3 ? amax + ((size_t)(2 * i) * M + m0) * CTN_AMAX_SLOTS : nullptr;        // max |x_in|, max |d| of this block
                unsigned* const ad = h3 ? amax + ((size_t)(2 * i + 1) * M + m0) * CTN_AMAX_SLOTS : nullptr;
                if (h3) {
                    if (step == 0)
                        rc = PROBED(F_K1, st, ctn_pw_gemm_h3(w1t, xin + xo, h1 + ho, Mc, H, B, K, Kp, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                             p[P_A1], s1, ax, nullptr, nullptr, st));
                    else if (step == 1)
                        rc = PROBED(F_K2, st, ctn_dw_fwd(h1 + ho, d + ho, p[P_D], Mc, H, K, Kp, P, dilation[i], causal, s1, w.np1, p[P_G1], p[P_B1], p[P_A1],
                                                         ms1 + 2 * m0, p[P_A2], s2, ad, st));
                    else
                        rc = PROBED(F_K3, st, ctn_pw_gemm_h3(w2t, d + ho, out + xo, Mc, B, H, K, Kp, s2, H, p[P_G2], p[P_B2], p[P_A2], ms2 + 2 * m0, xin + xo,
                                                             nullptr, nullptr, ad, gb + 2 * i, i + 1 < nblocks ? amax + ((size_t)(2 * i + 2) * M + m0) * CTN_AMAX_SLOTS : nullptr, st));
                } else if (step == 0)
                    rc = PROBED(F_K1, st, ctn_pw_gemm(w1t, xin + xo, h1 + ho, Mc, H, B, K, Kp, tw1, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                      p[P_A1], s1, 0, st));
                else if (step == 1)
                    rc = PROBED(F_K2, st, ctn_dw_fwd(h1 + ho, d + ho, p[P_D], Mc, H, K, Kp, P, dilation[i], causal, s1, w.np1, p[P_G1], p[P_B1], p[P_A1],
                                                     ms1 + 2 * m0, p[P_A2], s2, nullptr, st));
                else
                    rc = PROBED(F_K3, st, ctn_pw_gemm(w2t, d + ho, out + xo, Mc, B, H, K, Kp, tw2, s2, H, p[P_G2], p[P_B2], p[P_A2], ms2 + 2 * m0, xin + xo,
                                                      nullptr, nullptr, 0, st));
                if (rc) return rc;
            }
    }
    if (nch == 2 && (rc = ctn_stream_order(side_stream, stream))) return rc;
    return CTN_OK;
}

int ctn_tcn_gln_bwd(const void* const* params, void* const* grads, const int* dilation, int nblocks,
                    const float* x0, const float* xs, const float* h1s, const float* ds, const float* ms, const unsigned* amax,
                    const float* dout, float* dxs, float* dn1s,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream, int flags) {
    CTN_REQUIRE(params && grads && dilation && nblocks > 0 && x0 && xs && h1s && ds && ms && dout && dxs && dn1s && workspace,
                "ctn_tcn_gln_bwd: null pointer");
    const bool h3 = use_h3(B, H);
    CTN_REQUIRE(!h3 || amax, "ctn_tcn_gln_bwd: the h3 arithmetic needs the forward pass's amax array");
    CTN_REQUIRE(M > 0 && B > 0 && H > 0 && K > 0 && Kp >= K && P >= 1, "ctn_tcn_gln_bwd: bad sizes");
    const BwdWs w = bwd_ws(M, B, H, Kp, P, nblocks);
    if (workspace_bytes < w.total) {
        ctn_set_error("ctn_tcn_gln_bwd: workspace too small (%zu < %zu)", workspace_bytes, w.total);
        return CTN_ERR_WORKSPACE;
    }
    char* const ws = (char*)workspace;
    float* const dn2 = (float*)(ws + w.dn2);
    double* const s2p = (double*)(ws + w.s2p);
    // chained weight gradients (ctn_common.h): each launch's slabs are summed inside the next launch of the weight-gradient stream
    void* const slabs[2] = {ws + w.slab, ws + w.slab + align256(w.slab_bytes)};
    CtnWgradChain chain;
    int nwg = 0;
    const size_t xsz = (size_t)M * B * Kp, hsz = (size_t)M * H * Kp;
    void* const wst = side_stream ? side_stream : stream;       // where the weight gradients (and parameter-gradient sums) go
    int rc;
    for (int i = 0; i < nblocks; ++i)
        for (int j = 0; j < NPARAM; ++j)
            CTN_REQUIRE(params[(size_t)i * NPARAM + j] && grads[(size_t)i * NPARAM + j], "ctn_tcn_gln_bwd: block %d parameter / gradient %d is null", i, j);
    // b3 arithmetic: the transposed weight operands of both input-gradient GEMMs as bf16 pieces, two launches for the stack
    char* const wreg = ws + w.wp;
    const size_t slot = wslot_bytes(B, H);
    int twh = -1, twb = -1;
    float* const gb = (float*)(ws + w.gb);
    unsigned* const amax_b = (unsigned*)(ws + w.amax);          // [nblocks][dy | dh1][M]
    if ((rc = prepare_weights(params, nblocks, B, H, true, wreg, slot, &twh, &twb, h3, gb, stream))) return rc;
    if (h3) {
        if (hipMemsetAsync(amax_b, 0, (size_t)nblocks * 2 * M * CTN_AMAX_SLOTS * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) {
            ctn_set_error("ctn_tcn_gln_bwd: hipMemsetAsync failed");
            return CTN_ERR_LAUNCH;
        }
        if ((rc = PROBED(F_PREP, stream, ctn_absmax_rows(dout, M, (long long)B * Kp, amax_b + (size_t)(2 * (nblocks - 1)) * M * CTN_AMAX_SLOTS, stream)))) return rc;
    }
    const bool fuse4 = ctn_gln_fuse() != 0;       // no gLN-1' / PReLU-1' pass: its sums come out of B1's epilogue, B3 applies it
    const bool one_event = side_stream != nullptr && bwd_events(1) == 1;
    // the second 1x1 conv's weight gradient of block j: dW2 = dy_j . gLN2(prelu(d_j))^T -- it needs dy_j and forward tensors only
    auto wgrad2 = [&](int j) -> int {
        const float* const* pj = (const float* const*)(params + (size_t)j * NPARAM);
        float* const* gj = (float* const*)(grads + (size_t)j * NPARAM);
        const float* const dj = ds + (size_t)j * hsz;
        const float* const dyj = j == nblocks - 1 ? dout : dxs + (size_t)(j + 1) * xsz;
        const float* const ms2j = ms + ((size_t)j * 2 + 1) * M * 2;
        const unsigned* const adj = h3 ? amax + (size_t)(2 * j + 1) * M * CTN_AMAX_SLOTS : nullptr;
        unsigned* const adyj = amax_b + (size_t)(2 * j) * M * CTN_AMAX_SLOTS;
        if (h3) return PROBED(F_B2, wst, ctn_pw_wgrad_h3_chained(dyj, dj, gj[P_W2], M, B, H, K, Kp, pj[P_G2], pj[P_B2], pj[P_A2], ms2j, adyj, adj, gb + 2 * j, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
        return PROBED(F_B2, wst, ctn_pw_wgrad_chained(dyj, dj, gj[P_W2], M, B, H, K, Kp, pj[P_G2], pj[P_B2], pj[P_A2], ms2j, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
    };
    if (one_event) {        // dout, its maximum and the norm parameters' maxima are ready in stream order: the last block's dW2 can start
        if ((rc = ctn_stream_order(stream, side_stream))) return rc;
        if ((rc = wgrad2(nblocks - 1))) return rc;
    }
    for (int i = nblocks - 1; i >= 0; --i) {
        const float* const* p = (const float* const*)(params + (size_t)i * NPARAM);
        float* const* g = (float* const*)(grads + (size_t)i * NPARAM);
        const float* const x = i == 0 ? x0 : xs + (size_t)(i - 1) * xsz;
        const float* const h1 = h1s + (size_t)i * hsz;
        const float* const d = ds + (size_t)i * hsz;
        const unsigned* const ax = h3 ? amax + (size_t)(2 * i) * M * CTN_AMAX_SLOTS : nullptr;           // forward: max |x_in|, max |d|
        const unsigned* const ad = h3 ? amax + (size_t)(2 * i + 1) * M * CTN_AMAX_SLOTS : nullptr;
        unsigned* const ady = amax_b + (size_t)(2 * i) * M * CTN_AMAX_SLOTS;                             // backward: max |dy|, max |dh1|
        unsigned* const adh = amax_b + (size_t)(2 * i + 1) * M * CTN_AMAX_SLOTS;
        const float* const ms1 = ms + ((size_t)i * 2 + 0) * M * 2;
        const float* const ms2 = ms + ((size_t)i * 2 + 1) * M * 2;
        const float* const dy = i == nblocks - 1 ? dout : dxs + (size_t)(i + 1) * xsz;   // gradient of this block's output
        float* const dx = dxs + (size_t)i * xsz;
        float* const dn1 = dn1s + (size_t)i * hsz;      // a slot per block: the side stream still reads it while the chain moves on
        float* const pc = (float*)(ws + w.pc + (size_t)i * w.pc_slot);
        float* const da1p = (float*)(ws + w.da1p + (size_t)i * w.da1p_slot);
        double* const s1p = (double*)(ws + w.s1p + (size_t)i * w.s1p_slot);
        // second 1x1: input gradient (+ gLN2 backward sums); its weight gradient on the side stream
        if (fuse4) rc = PROBED(F_B1, stream, ctn_pw_dgrad_gln2(twh == -1 ? (const void*)p[P_W2] : (const void*)(wreg + (size_t)(2 * i) * slot), h3 ? 3 : (twh == 2 ? 2 : 1),
                               dy, dn2, M, H, B, K, Kp, d, p[P_G2], p[P_A2], ms2, p[P_G1], p[P_B1], p[P_D], P, dilation[i], causal, s2p, h3 ? ady : nullptr, stream));
        else if (h3) rc = PROBED(F_B1, stream, ctn_pw_dgrad_gln_h3(wreg + (size_t)(2 * i) * slot, dy, dn2, M, H, B, K, Kp, d, p[P_G2], p[P_A2], ms2, s2p, ady, stream));
        else if (twh == 2) rc = PROBED(F_B1, stream, ctn_pw_dgrad_gln_planes(wreg + (size_t)(2 * i) * slot, dy, dn2, M, H, B, K, Kp, d, p[P_G2], p[P_A2], ms2, s2p, stream));
        else rc = PROBED(F_B1, stream, ctn_pw_dgrad_gln(p[P_W2], dy, dn2, M, H, B, K, Kp, d, p[P_G2], p[P_A2], ms2, s2p, stream));
        if (rc) return rc;
        if (!one_event) {
            if (side_stream && (rc = ctn_stream_order(stream, side_stream))) return rc;
            if ((rc = wgrad2(i))) return rc;
        }
        // gLN2 <- PReLU2 <- depthwise <- gLN1 output in one pass, then gLN1 + PReLU1 backward in place
        if (fuse4) rc = PROBED(F_B3, stream, ctn_dw_bwd_gln2(dn2, d, h1, dn1, p[P_D], M, H, K, Kp, P, dilation[i], causal, p[P_G1], p[P_B1], p[P_A1], ms1,
                               p[P_G2], p[P_A2], ms2, s2p, w.np2, pc, h3 ? adh : nullptr, stream));       // (dn1 receives dh1)
        else rc = PROBED(F_B3, stream, ctn_dw_bwd(dn2, d, h1, dn1, p[P_D], M, H, K, Kp, P, dilation[i], causal, 1, p[P_G1], p[P_B1], p[P_A1], ms1,
                        p[P_G2], p[P_A2], ms2, s2p, w.np2, pc, s1p, stream));
        if (rc) return rc;
        // gLN1' / PReLU1' backward in place (B4).  (Folding it into the operand prologues of its two consumers was built and
        // measured in round 2: 10.23 vs 10.13 ms per step with B4 as its own pass -- both GEMMs then read h1 as well.)
        const int n_da1 = M * H;
        if (fuse4) rc = CTN_OK;
        else if (!(g_ctn_exp_skip & 1)) rc = PROBED(F_B4, stream, ctn_gln_prelu_bwd(dn1, h1, dn1, M, H, K, Kp, p[P_G1], p[P_A1], ms1, s1p, H, da1p, h3 ? adh : nullptr, stream));
        else if (h3) rc = ctn_absmax_rows(dn1, M, (long long)H * Kp, adh, stream);        // (lab: keep the h3 scales finite)
        if (rc) return rc;
        // first 1x1; the weight gradient and the fixed-order sums of this block's parameter-gradient partials feed only the
        // optimiser: second stream
        if (side_stream && !one_event && (rc = ctn_stream_order(stream, side_stream))) return rc;
        auto wgrad1 = [&]() -> int {
            if (h3) return PROBED(F_B6, wst, ctn_pw_wgrad_h3_chained(dn1, x, g[P_W1], M, H, B, K, Kp, nullptr, nullptr, nullptr, nullptr, adh, ax, nullptr, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
            return PROBED(F_B6, wst, ctn_pw_wgrad_chained(dn1, x, g[P_W1], M, H, B, K, Kp, nullptr, nullptr, nullptr, nullptr, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
        };
        auto finalize = [&](void* st) -> int {
            return PROBED(F_FIN, st, ctn_dw_bwd_finalize(pc, P, M, H, g[P_D], g[P_G2], g[P_B2], g[P_G1], g[P_B1], g[P_A2],
                                                        fuse4 ? pc + (size_t)(P + 5) * M * H : da1p, n_da1, g[P_A1], st));
        };
        if (side_stream && !one_event) {          // the fixed-order sums feed only the optimiser: second stream, behind the weight gradient
            if ((rc = wgrad1())) return rc;
            if ((rc = finalize(wst))) return rc;
        }
        if (h3) rc = PROBED(F_B5, stream, ctn_pw_gemm_h3(wreg + (size_t)(2 * i + 1) * slot, dn1, dx, M, B, H, K, Kp, nullptr, 0, nullptr, nullptr, nullptr, nullptr, dy,
                         nullptr, nullptr, adh, nullptr, i > 0 ? amax_b + (size_t)(2 * i - 2) * M * CTN_AMAX_SLOTS : nullptr, stream));
        else rc = PROBED(F_B5, stream, ctn_pw_gemm(twb == 2 ? (const float*)(wreg + (size_t)(2 * i + 1) * slot) : p[P_W1], dn1, dx, M, B, H, K, Kp, twb == 2 ? 2 : 1,
                         nullptr, 0, nullptr, nullptr, nullptr, nullptr, dy, nullptr, nullptr, 0, stream));
        if (rc) return rc;
        if (one_event) {            // one fork per block, behind B5: this block's dW1 and sums, then the next block's dW2 (its dy is this B5's output)
            if ((rc = ctn_stream_order(stream, side_stream))) return rc;
            if ((rc = wgrad1())) return rc;
            if ((rc = finalize(wst))) return rc;
            if (i > 0 && (rc = wgrad2(i - 1))) return rc;
        }
        if (!side_stream) {
            if ((rc = wgrad1())) return rc;
            if ((rc = finalize(stream))) return rc;
        }
    }
    if (chain.slab && (rc = PROBED(F_WFLUSH, wst, ctn_wgrad_chain_flush(&chain, wst)))) return rc;     // the last weight gradient's slabs
    // flags bit 0: leave the second stream un-joined (the caller issues more work behind it -- e.g. this bucket's gradient
    // all-reduce -- and joins later; it must then give every un-joined call a workspace of its own)
    if (side_stream && !(flags & 1) && (rc = ctn_stream_order(side_stream, stream))) return rc;
    return CTN_OK;
}

}  // extern "C"

// ---- cLN stack (causal BASELINE config): the same host-side composite over the un-fused norm kernels ------------------
namespace {
struct ClnBwdWs {
    size_t dn2, dd, dn1, pcw, pcn, dap, slab, wp, amax, colp, fc, total, slab_bytes, pcw_slot, pcn_slot, dap_slot;
    int ncol;
};
// form of the weight operand that ctn_pw_dgrad_cln gets from prepare_weights(backward): 3 h3 pieces, 2 b6 pieces, 1 the stored matrix
inline int cln_w_form(bool h3, int twh) { return h3 ? 3 : (twh == 2 ? 2 : 1); }
ClnBwdWs cln_bwd_ws(int M, int B, int H, int Kp, int P, int nblocks) {
    ClnBwdWs w;
    size_t o = 0;
    const size_t hsz = align256((size_t)M * H * Kp * sizeof(float));
    w.dn2 = o; o += hsz;
    w.dd = o; o += hsz;
    w.dn1 = o; o += hsz;
    // parameter-gradient partials: a slot per block (two per block for the norms) -- their fixed-order sums run on the
    // weight-gradient stream while the chain is already in the next block
    w.pcw_slot = align256((size_t)(P + 3) * M * H * sizeof(float));       // (fused second norm: taps + dgamma2, dbeta2, dalpha2 partials)
    w.pcn_slot = align256(ctn_cln_bwd_pc_floats(M, H, Kp) * sizeof(float));
    w.dap_slot = align256((size_t)ctn_cln_bwd_blocks(M, Kp) * sizeof(float));
    w.pcw = o; o += (size_t)nblocks * w.pcw_slot;
    w.pcn = o; o += (size_t)nblocks * 2 * w.pcn_slot;
    w.dap = o; o += (size_t)nblocks * 2 * w.dap_slot;
    const size_t s1 = ctn_pw_wgrad_workspace(M, H, B, Kp), s2 = ctn_pw_wgrad_workspace(M, B, H, Kp);
    w.slab_bytes = s1 > s2 ? s1 : s2;
    w.slab = o; o += 2 * align256(w.slab_bytes);          // two: a chained weight gradient writes one while the next launch sums the other
    w.wp = o; o += (size_t)nblocks * 2 * wslot_bytes(B, H);
    w.amax = o; o += align256((size_t)nblocks * 2 * M * CTN_AMAX_SLOTS * sizeof(unsigned));     // h3: tracked maxima of (dy, dh1) per block
    // fused second norm (ctn_tune("cln_fuse")): per-frame column partials of the input-gradient GEMM and the per-frame constants
    const bool h3 = use_h3(B, H);
    w.ncol = ctn_pw_col_parts(M, H, Kp, h3 ? 3 : (pieces(H) ? 2 : 1));
    w.colp = o; o += align256((size_t)M * w.ncol * Kp * 2 * sizeof(double));
    w.fc = o; o += align256((size_t)M * 4 * Kp * sizeof(float));
    w.total = o;
    return w;
}
}  // namespace

extern "C" int ctn_cln_fuse(void);       // csrc/ctn_tcn.hip

extern "C" {

// forward weight-operand form of ctn_pw_gemm_cln after prepare_weights(forward): 3 h3 pieces, 2 b6 pieces, 1 the [I, O] fp32 copy
static int cln_fwd_w_form(int B, int H) { return use_h3(B, H) ? 3 : (pieces(H) ? 2 : 1); }
size_t ctn_tcn_cln_fwd_workspace(int M, int B, int H, int Kp, int nblocks) {
    // [nblocks][w1 operand | w2 operand], then the column partials of the first 1x1 conv (ctn_tune("cln_fuse", 2))
    return align256((size_t)nblocks * 2 * wslot_bytes(B, H)) +
           align256((size_t)M * ctn_pw_col_parts(M, H, Kp, cln_fwd_w_form(B, H)) * Kp * 2 * sizeof(double));
}
size_t ctn_tcn_cln_bwd_workspace(int M, int B, int H, int Kp, int P, int nblocks) { return cln_bwd_ws(M, B, H, Kp, P, nblocks).total; }

int ctn_tcn_cln_fwd(const void* const* params, const int* dilation, int nblocks, const float* x0,
                    float* xs, float* h1s, float* n1s, float* ds, float* n2s, float* st, unsigned* amax, int save,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream) {
    CTN_REQUIRE(params && dilation && nblocks > 0 && x0 && xs && h1s && n1s && ds && n2s && st && workspace, "ctn_tcn_cln_fwd: null pointer");
    const bool h3 = use_h3(B, H);
    CTN_REQUIRE(!h3 || amax, "ctn_tcn_cln_fwd: the h3 arithmetic needs the amax array");
    CTN_REQUIRE(M > 0 && B > 0 && H > 0 && K > 0 && Kp >= K && P >= 1, "ctn_tcn_cln_fwd: bad sizes");
    if (workspace_bytes < ctn_tcn_cln_fwd_workspace(M, B, H, Kp, nblocks)) {
        ctn_set_error("ctn_tcn_cln_fwd: workspace too small (%zu < %zu)", workspace_bytes, ctn_tcn_cln_fwd_workspace(M, B, H, Kp, nblocks));
        return CTN_ERR_WORKSPACE;
    }
    const size_t xsz = (size_t)M * B * Kp, hsz = (size_t)M * H * Kp, ssz = (size_t)M * Kp;
    char* const wreg = (char*)workspace;
    const size_t slot = wslot_bytes(B, H);
    int rc, tw1 = 0, tw2 = 0;
    for (int i = 0; i < nblocks; ++i)
        for (int j = 0; j < NPARAM; ++j) CTN_REQUIRE(params[(size_t)i * NPARAM + j], "ctn_tcn_cln_fwd: block %d parameter %d is null", i, j);
    if ((rc = prepare_weights(params, nblocks, B, H, false, wreg, slot, &tw1, &tw2, h3, nullptr, stream))) return rc;
    // ctn_tune("cln_fuse", 2): the first norm has no pass and no stored output -- its per-frame statistics come out of K1's epilogue
    // (column partials + ctn_cln_stats_frame) and the norm is applied in the depthwise kernel's prologue
    const bool fuse1 = ctn_cln_fuse() >= 2;
    const int wform = cln_fwd_w_form(B, H), ncol = ctn_pw_col_parts(M, H, Kp, wform);
    double* const colp = (double*)((char*)workspace + align256((size_t)nblocks * 2 * wslot_bytes(B, H)));
    if (h3) {
        if (hipMemsetAsync(amax, 0, (size_t)nblocks * 2 * M * CTN_AMAX_SLOTS * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) {
            ctn_set_error("ctn_tcn_cln_fwd: hipMemsetAsync failed");
            return CTN_ERR_LAUNCH;
        }
        if ((rc = PROBED(F_PREP, stream, ctn_absmax_rows(x0, M, (long long)B * Kp, amax, stream)))) return rc;
    }
    // two half-batch chains on two streams, as in ctn_tcn_gln_fwd (cLN statistics are per frame of one utterance)
    const int nch = (side_stream != nullptr && M >= 2) ? 2 : 1;
    const int m0c[2] = {0, M / 2}, mcc[2] = {nch == 2 ? M / 2 : M, M - M / 2};
    void* const sts[2] = {stream, side_stream};
    if (nch == 2 && (rc = ctn_stream_order(stream, side_stream))) return rc;
    for (int i = 0; i < nblocks; ++i) {
        const float* const* p = (const float* const*)(params + (size_t)i * NPARAM);
        const size_t s = save ? (size_t)i : 0;
        const float* const xin = i == 0 ? x0 : xs + (save ? (size_t)(i - 1) : (size_t)((i - 1) & 1)) * xsz;
        float* const h1 = h1s + s * hsz; float* const n1 = n1s + s * hsz; float* const d = ds + s * hsz; float* const n2 = n2s + s * hsz;
        float* const out = xs + (save ? (size_t)i : (size_t)(i & 1)) * xsz;
        float* const stb = st + s * 4 * ssz;
        for (int step = 0; step < 5; ++step)
            for (int c = 0; c < nch; ++c) {
                const int m0 = m0c[c], Mc = mcc[c];
                void* const sc = sts[c];
                const size_t xo = (size_t)m0 * B * Kp, ho = (size_t)m0 * H * Kp, so = (size_t)m0 * Kp;
                unsigned* const ax = h3 ? amax + ((size_t)(2 * i) * M + m0) * CTN_AMAX_SLOTS : nullptr;        // max |x_in|, max |n2| of this block
                unsigned* const an = h3 ? amax + ((size_t)(2 * i + 1) * M + m0) * CTN_AMAX_SLOTS : nullptr;
                double* const cp = colp + (size_t)m0 * ncol * Kp * 2;
                if (step == 0 && fuse1)
                    rc = PROBED(F_K1, sc, ctn_pw_gemm_cln(wreg + (size_t)(2 * i) * slot, wform, xin + xo, h1 + ho, Mc, H, B, K, Kp, p[P_A1], cp, ax, sc));
                else if (step == 1 && fuse1)
                    rc = PROBED(F_FRAME, sc, ctn_cln_stats_frame(cp, ncol, stb + so, stb + ssz + so, Mc, H, Kp, sc));
                else if (step == 2 && fuse1)
                    rc = PROBED(F_K2, sc, ctn_dw_fwd_cln(h1 + ho, d + ho, p[P_D], Mc, H, K, Kp, P, dilation[i], causal, stb + so, stb + ssz + so,
                                                         p[P_G1], p[P_B1], p[P_A1], sc));
                else if (step == 0 && h3)
                    rc = PROBED(F_K1, sc, ctn_pw_gemm_h3(wreg + (size_t)(2 * i) * slot, xin + xo, h1 + ho, Mc, H, B, K, Kp, nullptr, 0, nullptr, nullptr, nullptr,
                                                         nullptr, nullptr, nullptr, nullptr, ax, nullptr, nullptr, sc));
                else if (step == 0)
                    rc = PROBED(F_K1, sc, ctn_pw_gemm((const float*)(wreg + (size_t)(2 * i) * slot), xin + xo, h1 + ho, Mc, H, B, K, Kp, tw1, nullptr, 0, nullptr,
                                                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, sc));
                else if (step == 1)
                    rc = PROBED(F_CLN_FWD, sc, ctn_cln_fwd(h1 + ho, n1 + ho, stb + so, stb + ssz + so, Mc, H, K, Kp, p[P_G1], p[P_B1], p[P_A1], nullptr, sc));
                else if (step == 2)
                    rc = PROBED(F_K2, sc, ctn_dw_fwd(n1 + ho, d + ho, p[P_D], Mc, H, K, Kp, P, dilation[i], causal, nullptr, 0, nullptr, nullptr, nullptr,
                                                     nullptr, nullptr, nullptr, nullptr, sc));
                else if (step == 3)
                    rc = PROBED(F_CLN_FWD, sc, ctn_cln_fwd(d + ho, n2 + ho, stb + 2 * ssz + so, stb + 3 * ssz + so, Mc, H, K, Kp, p[P_G2], p[P_B2], p[P_A2], an, sc));
                else if (h3)
                    rc = PROBED(F_K3, sc, ctn_pw_gemm_h3(wreg + (size_t)(2 * i + 1) * slot, n2 + ho, out + xo, Mc, B, H, K, Kp, nullptr, 0, nullptr, nullptr, nullptr,
                                                         nullptr, xin + xo, nullptr, nullptr, an, nullptr,
                                                         i + 1 < nblocks ? amax + ((size_t)(2 * i + 2) * M + m0) * CTN_AMAX_SLOTS : nullptr, sc));
                else
                    rc = PROBED(F_K3, sc, ctn_pw_gemm((const float*)(wreg + (size_t)(2 * i + 1) * slot), n2 + ho, out + xo, Mc, B, H, K, Kp, tw2, nullptr, 0,
                                                      nullptr, nullptr, nullptr, nullptr, xin + xo, nullptr, nullptr, 0, sc));
                if (rc) return rc;
            }
    }
    if (nch == 2 && (rc = ctn_stream_order(side_stream, stream))) return rc;
    return CTN_OK;
}

int ctn_tcn_cln_bwd(const void* const* params, void* const* grads, const int* dilation, int nblocks,
                    const float* x0, const float* xs, const float* h1s, const float* n1s, const float* ds, const float* n2s,
                    const float* st, const unsigned* amax, const float* dout, float* dxs, float* dh1s,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream, int flags) {
    CTN_REQUIRE(params && grads && dilation && nblocks > 0 && x0 && xs && h1s && n1s && ds && n2s && st && dout && dxs && dh1s && workspace,
                "ctn_tcn_cln_bwd: null pointer");
    const bool h3 = use_h3(B, H);
    CTN_REQUIRE(!h3 || amax, "ctn_tcn_cln_bwd: the h3 arithmetic needs the forward pass's amax array");
    CTN_REQUIRE(M > 0 && B > 0 && H > 0 && K > 0 && Kp >= K && P >= 1, "ctn_tcn_cln_bwd: bad sizes");
    const ClnBwdWs w = cln_bwd_ws(M, B, H, Kp, P, nblocks);
    if (workspace_bytes < w.total) {
        ctn_set_error("ctn_tcn_cln_bwd: workspace too small (%zu < %zu)", workspace_bytes, w.total);
        return CTN_ERR_WORKSPACE;
    }
    char* const ws = (char*)workspace;
    float* const dn2 = (float*)(ws + w.dn2);
    float* const dd = (float*)(ws + w.dd);
    float* const dn1 = (float*)(ws + w.dn1);
    double* const colp = (double*)(ws + w.colp);
    float* const fc = (float*)(ws + w.fc);
    const bool fuse = ctn_cln_fuse() != 0, fuse1 = ctn_cln_fuse() >= 2;      // fuse1: n1 was never stored (forward under the same setting)
    // chained weight gradients (ctn_common.h): each launch's slabs are summed inside the next launch of the weight-gradient stream
    void* const slabs[2] = {ws + w.slab, ws + w.slab + align256(w.slab_bytes)};
    CtnWgradChain chain;
    int nwg = 0;
    const size_t xsz = (size_t)M * B * Kp, hsz = (size_t)M * H * Kp, ssz = (size_t)M * Kp;
    void* const wst = side_stream ? side_stream : stream;
    int rc;
    for (int i = 0; i < nblocks; ++i)
        for (int j = 0; j < NPARAM; ++j)
            CTN_REQUIRE(params[(size_t)i * NPARAM + j] && grads[(size_t)i * NPARAM + j], "ctn_tcn_cln_bwd: block %d parameter / gradient %d is null", i, j);
    char* const wreg = ws + w.wp;
    const size_t slot = wslot_bytes(B, H);
    int twh = -1, twb = -1;
    if ((rc = prepare_weights(params, nblocks, B, H, true, wreg, slot, &twh, &twb, h3, nullptr, stream))) return rc;
    unsigned* const amax_b = (unsigned*)(ws + w.amax);          // [nblocks][dy | dh1][M][slots]
    if (h3) {
        if (hipMemsetAsync(amax_b, 0, (size_t)nblocks * 2 * M * CTN_AMAX_SLOTS * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) {
            ctn_set_error("ctn_tcn_cln_bwd: hipMemsetAsync failed");
            return CTN_ERR_LAUNCH;
        }
        if ((rc = PROBED(F_PREP, stream, ctn_absmax_rows(dout, M, (long long)B * Kp, amax_b + (size_t)(2 * (nblocks - 1)) * M * CTN_AMAX_SLOTS, stream)))) return rc;
    }
    const bool one_event = side_stream != nullptr && bwd_events(2) == 1;       // (see ctn_tcn_gln_bwd)
    auto wgrad2 = [&](int j) -> int {       // dW2 of block j = dy_j . n2_j^T: needs dy_j and forward tensors only
        float* const* gj = (float* const*)(grads + (size_t)j * NPARAM);
        const float* const n2j = n2s + (size_t)j * hsz;
        const float* const dyj = j == nblocks - 1 ? dout : dxs + (size_t)(j + 1) * xsz;
        const unsigned* const anj = h3 ? amax + (size_t)(2 * j + 1) * M * CTN_AMAX_SLOTS : nullptr;
        unsigned* const adyj = amax_b + (size_t)(2 * j) * M * CTN_AMAX_SLOTS;
        if (h3) return PROBED(F_B2, wst, ctn_pw_wgrad_h3_chained(dyj, n2j, gj[P_W2], M, B, H, K, Kp, nullptr, nullptr, nullptr, nullptr, adyj, anj, nullptr, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
        return PROBED(F_B2, wst, ctn_pw_wgrad_chained(dyj, n2j, gj[P_W2], M, B, H, K, Kp, nullptr, nullptr, nullptr, nullptr, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
    };
    if (one_event) {
        if ((rc = ctn_stream_order(stream, side_stream))) return rc;
        if ((rc = wgrad2(nblocks - 1))) return rc;
    }
    for (int i = nblocks - 1; i >= 0; --i) {
        const float* const* p = (const float* const*)(params + (size_t)i * NPARAM);
        float* const* g = (float* const*)(grads + (size_t)i * NPARAM);
        const float* const x = i == 0 ? x0 : xs + (size_t)(i - 1) * xsz;
        const unsigned* const ax = h3 ? amax + (size_t)(2 * i) * M * CTN_AMAX_SLOTS : nullptr;           // forward: max |x_in|, max |n2|
        const unsigned* const an = h3 ? amax + (size_t)(2 * i + 1) * M * CTN_AMAX_SLOTS : nullptr;
        unsigned* const ady = amax_b + (size_t)(2 * i) * M * CTN_AMAX_SLOTS;                             // backward: max |dy|, max |dh1|
        unsigned* const adh = amax_b + (size_t)(2 * i + 1) * M * CTN_AMAX_SLOTS;
        const float* const h1 = h1s + (size_t)i * hsz; const float* const n1 = n1s + (size_t)i * hsz;
        const float* const d = ds + (size_t)i * hsz; const float* const n2 = n2s + (size_t)i * hsz;
        const float* const stb = st + (size_t)i * 4 * ssz;
        const float* const dy = i == nblocks - 1 ? dout : dxs + (size_t)(i + 1) * xsz;
        float* const dx = dxs + (size_t)i * xsz;
        float* const dh1 = dh1s + (size_t)i * hsz;          // a slot per block: the side stream reads it while the chain moves on
        float* const pcw = (float*)(ws + w.pcw + (size_t)i * w.pcw_slot);
        float* const pcn2 = (float*)(ws + w.pcn + (size_t)(2 * i) * w.pcn_slot), * const pcn1 = (float*)(ws + w.pcn + (size_t)(2 * i + 1) * w.pcn_slot);
        float* const dap2 = (float*)(ws + w.dap + (size_t)(2 * i) * w.dap_slot), * const dap1 = (float*)(ws + w.dap + (size_t)(2 * i + 1) * w.dap_slot);
        // fused second norm: the GEMM's epilogue also reads d and leaves the per-frame sums over channels of gamma2 dn2 and
        // gamma2 dn2 xhat2 as column partials; the stand-alone cln_bwd pass (dn2, d -> dd: three tensor passes) is gone
        if (fuse) rc = PROBED(F_B1, stream, ctn_pw_dgrad_cln(twh == -1 ? (const void*)p[P_W2] : (const void*)(wreg + (size_t)(2 * i) * slot), cln_w_form(h3, twh), dy, dn2,
                              M, H, B, K, Kp, d, p[P_G2], p[P_A2], stb + 2 * ssz, stb + 3 * ssz, colp, h3 ? ady : nullptr, stream));
        else if (h3) rc = PROBED(F_B1, stream, ctn_pw_gemm_h3(wreg + (size_t)(2 * i) * slot, dy, dn2, M, H, B, K, Kp, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, nullptr, ady, nullptr, nullptr, stream));
        else rc = PROBED(F_B1, stream, ctn_pw_gemm(twh == 2 ? (const float*)(wreg + (size_t)(2 * i) * slot) : p[P_W2], dy, dn2, M, H, B, K, Kp, twh == 2 ? 2 : 1,
                              nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream));
        if (rc) return rc;
        if (!one_event) {
            if (side_stream && (rc = ctn_stream_order(stream, side_stream))) return rc;
            if ((rc = wgrad2(i))) return rc;
        }
        if (fuse) {
            if ((rc = PROBED(F_FRAME, stream, ctn_cln_bwd_frame(colp, w.ncol, stb + 2 * ssz, stb + 3 * ssz, fc, M, H, Kp, stream)))) return rc;
            if (fuse1) rc = PROBED(F_B3, stream, ctn_dw_bwd_cln(dn2, d, h1, dn1, p[P_D], M, H, K, Kp, P, dilation[i], causal, p[P_G2], p[P_A2], fc,
                                                                p[P_G1], p[P_B1], p[P_A1], stb, stb + ssz, pcw, stream));
            else rc = PROBED(F_B3, stream, ctn_dw_bwd_cln(dn2, d, n1, dn1, p[P_D], M, H, K, Kp, P, dilation[i], causal, p[P_G2], p[P_A2], fc,
                                                          nullptr, nullptr, nullptr, nullptr, nullptr, pcw, stream));
            if (rc) return rc;
        } else {
            if ((rc = PROBED(F_CLN_BWD, stream, ctn_cln_bwd(dn2, d, dd, stb + 2 * ssz, stb + 3 * ssz, M, H, K, Kp, p[P_G2], p[P_A2], nullptr, nullptr, dap2, pcn2, nullptr, stream)))) return rc;
            if ((rc = PROBED(F_B3, stream, ctn_dw_bwd(dd, nullptr, n1, dn1, p[P_D], M, H, K, Kp, P, dilation[i], causal, 0, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, 0, pcw, nullptr, stream)))) return rc;
        }
        if ((rc = PROBED(F_CLN_BWD, stream, ctn_cln_bwd(dn1, h1, dh1, stb, stb + ssz, M, H, K, Kp, p[P_G1], p[P_A1], nullptr, nullptr, dap1, pcn1, h3 ? adh : nullptr, stream)))) return rc;
        // the fixed-order parameter-gradient sums of this block feed only the optimiser: weight-gradient stream
        void* const fst = wst;
        auto fins = [&]() -> int {
            int r;
            if (fuse) {
                if ((r = PROBED(F_TAPS, fst, ctn_dw_bwd_cln_finalize(pcw, P, M, H, g[P_D], g[P_G2], g[P_B2], g[P_A2], fst)))) return r;
            } else {
                if ((r = PROBED(F_FIN, fst, ctn_cln_bwd_finalize(pcn2, dap2, M, H, Kp, g[P_G2], g[P_B2], g[P_A2], fst)))) return r;
                if ((r = PROBED(F_TAPS, fst, ctn_dw_bwd_taps(pcw, P, M, H, g[P_D], fst)))) return r;
            }
            return PROBED(F_FIN, fst, ctn_cln_bwd_finalize(pcn1, dap1, M, H, Kp, g[P_G1], g[P_B1], g[P_A1], fst));
        };
        auto wgrad1 = [&]() -> int {
            if (h3) return PROBED(F_B6, wst, ctn_pw_wgrad_h3_chained(dh1, x, g[P_W1], M, H, B, K, Kp, nullptr, nullptr, nullptr, nullptr, adh, ax, nullptr, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
            return PROBED(F_B6, wst, ctn_pw_wgrad_chained(dh1, x, g[P_W1], M, H, B, K, Kp, nullptr, nullptr, nullptr, nullptr, slabs[nwg++ & 1], w.slab_bytes, wst, &chain));
        };
        if (!one_event) {
            if (side_stream && (rc = ctn_stream_order(stream, side_stream))) return rc;
            if ((rc = fins())) return rc;
            if (side_stream && (rc = wgrad1())) return rc;
        }
        if (h3) rc = PROBED(F_B5, stream, ctn_pw_gemm_h3(wreg + (size_t)(2 * i + 1) * slot, dh1, dx, M, B, H, K, Kp, nullptr, 0, nullptr, nullptr, nullptr, nullptr, dy,
                              nullptr, nullptr, adh, nullptr, i > 0 ? amax_b + (size_t)(2 * i - 2) * M * CTN_AMAX_SLOTS : nullptr, stream));
        else rc = PROBED(F_B5, stream, ctn_pw_gemm(twb == 2 ? (const float*)(wreg + (size_t)(2 * i + 1) * slot) : p[P_W1], dh1, dx, M, B, H, K, Kp, twb == 2 ? 2 : 1,
                              nullptr, 0, nullptr, nullptr, nullptr, nullptr, dy, nullptr, nullptr, 0, stream));
        if (rc) return rc;
        if (one_event) {            // one fork per block, behind B5 (see ctn_tcn_gln_bwd)
            if ((rc = ctn_stream_order(stream, side_stream))) return rc;
            if ((rc = fins())) return rc;
            if ((rc = wgrad1())) return rc;
            if (i > 0 && (rc = wgrad2(i - 1))) return rc;       // (ahead of the sums instead: 12.09 vs 11.97 ms; two forks: 11.60)
        } else if (!side_stream && (rc = wgrad1())) return rc;
    }
    if (chain.slab && (rc = PROBED(F_WFLUSH, wst, ctn_wgrad_chain_flush(&chain, wst)))) return rc;     // the last weight gradient's slabs
    // flags bit 0: leave the second stream un-joined (the caller issues more work behind it -- e.g. this bucket's gradient
    // all-reduce -- and joins later; it must then give every un-joined call a workspace of its own)
    if (side_stream && !(flags & 1) && (rc = ctn_stream_order(side_stream, stream))) return rc;
    return CTN_OK;
}

}  // extern "C"
