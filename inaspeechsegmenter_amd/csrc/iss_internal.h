// Internal state of libiss_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>
#include <vector>
#include <unordered_map>
#include <map>
#include "../../include/iss.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct ConvOp;   // cnn.hip

struct IssNet {
    bool loaded = false;
    std::vector<int32_t> prog;            // nrows * ISS_PROG_COLS
    int nrows = 0;
    float* d_blob = nullptr;              // parameters (conv weights [Cout][Kpad])
    uint16_t* d_wh = nullptr;             // bf16 hi part of every blob element (same offsets)
    uint16_t* d_wl = nullptr;             // bf16 lo part
    uint16_t* d_wh16 = nullptr;           // fp16 hi / lo parts (ISS_PREC_F16X3)
    uint16_t* d_wl16 = nullptr;
    bool f16_ok = false;                  // every parameter is inside fp16's range
    std::map<std::pair<int, int>, uint16_t*> dhl_wp;   // (row, fp16?) -> the dense layer's weights packed for conv_dhl_kernel (built at first use)
    float* d_wsum = nullptr;              // patch-mode first layers: sum_k w[c][k] per output channel (shared first layer)
    std::vector<int64_t> wsum_off;        // per row: offset into d_wsum, -1 = none
    std::vector<int64_t> wsumx_off;       // per row: offset of S[W][Cout] (zero-padded first layers: per-column weight sums), -1 = none
    int64_t blob_floats = 0;
    std::vector<int64_t> w_dev_off;       // per row: offset of the padded weight matrix in d_blob
    std::vector<int32_t> kpad;            // per row: K padded to the GEMM k-tile
    int32_t* d_ktab = nullptr;            // per-conv im2col tables
    std::vector<int64_t> ktab_off;        // per row
    int nbuf = 0;
    std::vector<int64_t> buf_elems;
    int in_h = 0, in_w = 0, in_c = 0, out_dim = 0;
    double flops_per_sample = 0;
    std::unordered_map<long long, int> fp_pix;   // (row << 32 | samples) -> pixels a 128-row tile's LDS footprint spans
    // precision guard (iss_set_precision_guard / iss_cnn_precision_info)
    int prec_override = -1;               // -1: the context's mode; else ISS_PREC_* for this network only
    int guard_state = 0;                  // ISS_GUARD_*
    float guard_dlogp = -1.f;             // max |log p(mode asked for) - log p(exact f32)| of the probe, -1 = never probed
    float guard_dlogp_chosen = -1.f;      // the same figure for the mode the network runs in after the probe
    int guard_slots = 0;                  // windows the probe compared
};

struct iss_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;

    // sidekit tables
    bool sk_tables = false;
    double* d_window = nullptr;           // 400
    float* d_melw = nullptr;              // packed non-zero weights per filter
    int32_t* d_mellim = nullptr;          // 24 x {first_bin, n_bins, weight_offset}
    double* d_tw = nullptr;               // W256 (256 complex) then W512 (256 complex)

    // resident signal
    DevBuf sig;                           // owned copy
    const void* sig_ptr = nullptr;        // points into sig.p or to caller's device memory
    int sig_kind = 0;                     // 0 none, 1 pcm16, 2 f32
    int64_t sig_n = 0;

    // resident features
    DevBuf mspec, loge;
    int32_t T = 0;
    bool have_feats = false;

    // CNN engine
    IssNet nets[ISS_MAX_NETS];
    uint64_t ws_limit = 24ull << 30;                // of 288 GB: the x-vector net then runs the 1 816 windows per pass its 2^31-element buffers allow (12 GiB: 1 072, -2 %)
    int precision = ISS_PREC_F16X3;
    float guard_threshold = 5e-4f;        // precision guard: escalate a patch network to exact f32 above this max |d log p| (<= 0: guard off)
    bool in_guard = false;
    uint32_t diag = 0;                    // ISS_DIAG_* kernel-selection switches (iss_set_diag; 0 in production)
    std::vector<DevBuf> act;              // activation buffers (grown on demand)
    DevBuf d_winrow, d_stats, d_finite, d_out, d_in;
    DevBuf raw1;                          // shared first layer: raw conv over the log-mel rows of the current chunk

    // vbx
    bool vbx_tables = false;
    double* d_vbx_window = nullptr;
    double* d_vbx_melw = nullptr;
    int32_t* d_vbx_mellim = nullptr;
    DevBuf vbx_sig, vbx_dither, vbx_fb, vbx_out;
    int32_t vbx_T = 0;
    int64_t vbx_dither_n = 0;              // length of the dither stream cached in vbx_dither (0 = none)

    // pinned staging buffers of the asynchronous entry points (window lists): reused once their copy has completed
    struct Staging { void* p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
    std::vector<Staging> staging;
    // tickets of iss_cnn_probs_async: ticket t <-> ticket_ev[t - ticket_base]
    std::vector<hipEvent_t> ticket_ev;
    int64_t ticket_base = 1;
    hipEvent_t order_ev = nullptr;        // iss_signal_pcm16_device*: producer stream -> library stream

    // multi-GPU (RCCL, dlopen'ed)
    void* comm = nullptr;                 // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    DevBuf comm_send, comm_recv;

    // profiling
    bool prof = false;
    double prof_ms[ISS_PROF_KINDS] = {};
    int64_t prof_launch[ISS_PROF_KINDS] = {};
    double prof_flops[ISS_PROF_KINDS] = {};
    double prof_row_ms[ISS_PROF_ROWS] = {};
    int64_t prof_row_launch[ISS_PROF_ROWS] = {};
    struct Pending { hipEvent_t a, b; int kind; int sub; int row; double flops; std::string inst; };
    std::vector<Pending> pending;
    // per kernel INSTANTIATION (template arguments spelled out by the launcher): what bench.py's roofline.dominant reports
    struct Inst { std::string name; double ms = 0; int64_t launches = 0; double flops = 0; };
    std::vector<Inst> prof_inst;
    std::vector<hipEvent_t> ev_pool;
};

int iss_fail(iss_ctx* c, int code, const char* fmt, ...);
int iss_reserve(iss_ctx* c, DevBuf& b, size_t bytes);
int iss_stage_host(iss_ctx* c, const void* src, size_t bytes, void** pinned_out, int* slot_out);   // copy into a pinned staging buffer
void iss_stage_mark(iss_ctx* c, int slot);                                                       // record 'consumed' on the stream

#define ISS_HIP(c, call)                                                                  \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess)                                                            \
            return iss_fail((c), ISS_EHIP, "%s failed: %s (%s:%d)", #call,                \
                            hipGetErrorString(e__), __FILE__, __LINE__);                  \
    } while (0)

// profiling brackets: record events around a kernel class when enabled
void iss_prof_begin(iss_ctx* c, int kind, double flops);
void iss_prof_tag(iss_ctx* c, int sub);      // kernel class (ISS_PROF_* >= 3) of the launch bracketed last: counted there too
void iss_prof_row(iss_ctx* c, int row);      // op-program row of the launch bracketed last
void iss_prof_inst(iss_ctx* c, const char* fmt, ...);   // kernel instantiation of the launch bracketed last (printf-style name)
void iss_prof_end(iss_ctx* c);
void iss_prof_collect(iss_ctx* c);

// implemented in the .hip files
int iss_launch_sidekit(iss_ctx* c);
int iss_cnn_free(iss_ctx* c, int net_id);
