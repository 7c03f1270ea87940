// Context management, signal ingest, feature readback and profiling hooks of libiss_hip.so.
// Host code only (HIP runtime API); the kernels live in sidekit.hip / cnn.hip / vbx.hip.
#include "iss_internal.h"
#include <cstring>
#include <cmath>
#include <mutex>

static std::string g_err;   // creation-time errors (no context yet)
static std::mutex g_err_mu;

int iss_fail(iss_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else { std::lock_guard<std::mutex> l(g_err_mu); g_err = buf; }
    return code;
}

int iss_reserve(iss_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return ISS_OK;
    if (b.p) { ISS_HIP(c, hipStreamSynchronize(c->stream)); ISS_HIP(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + (bytes >> 3) + 256;       // a little slack so growing inputs do not thrash
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return iss_fail(c, ISS_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return ISS_OK;
}

extern "C" const char* iss_version(void) { return "iss-hip 0.1 (gfx950)"; }

extern "C" const char* iss_last_error(const iss_ctx* c) {
    if (c) return c->err.c_str();
    std::lock_guard<std::mutex> l(g_err_mu);
    return g_err.c_str();
}

extern "C" int iss_create(int device_id, iss_ctx** out) {
    if (!out) return iss_fail(nullptr, ISS_EINVAL, "iss_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return iss_fail(nullptr, ISS_ENODEV, "no HIP device visible (%s); this library has no CPU fallback",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return iss_fail(nullptr, ISS_EINVAL, "device_id %d out of range [0,%d)", device_id, ndev);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return iss_fail(nullptr, ISS_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return iss_fail(nullptr, ISS_ENODEV, "device %d is %s; this library is built for gfx950 only",
                        device_id, prop.gcnArchName);
    iss_ctx* c = new iss_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return iss_fail(nullptr, ISS_EHIP, "cannot create a stream on device %d", device_id);
    }
    *out = c;
    return ISS_OK;
}

static void free_buf(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

extern "C" void iss_destroy(iss_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < ISS_MAX_NETS; ++i) iss_cnn_free(c, i);
    void* singles[] = {c->d_window, c->d_melw, c->d_mellim, c->d_tw, c->d_vbx_window, c->d_vbx_melw, c->d_vbx_mellim};
    for (void* p : singles) if (p) (void)hipFree(p);
    DevBuf* bufs[] = {&c->sig, &c->mspec, &c->loge, &c->d_winrow, &c->d_stats, &c->d_finite, &c->d_out, &c->d_in, &c->raw1,
                      &c->vbx_sig, &c->vbx_dither, &c->vbx_fb, &c->vbx_out};
    for (DevBuf* b : bufs) free_buf(*b);
    for (auto& b : c->act) free_buf(b);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto& e : c->ticket_ev) (void)hipEventDestroy(e);
    for (auto& s : c->staging) { if (s.p) (void)hipHostFree(s.p); if (s.done) (void)hipEventDestroy(s.done); }
    if (c->order_ev) (void)hipEventDestroy(c->order_ev);
    iss_comm_destroy(c);
    free_buf(c->comm_send); free_buf(c->comm_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int iss_set_workspace_limit(iss_ctx* c, uint64_t bytes) {
    if (!c) return ISS_EINVAL;
    if (bytes < (64ull << 20)) return iss_fail(c, ISS_EINVAL, "workspace limit below 64 MiB");
    c->ws_limit = bytes;
    return ISS_OK;
}

extern "C" int iss_synchronize(iss_ctx* c) {
    if (!c) return ISS_EINVAL;
    ISS_HIP(c, hipSetDevice(c->device));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    return ISS_OK;
}

// ---------------------------------------------------------------- sidekit tables
extern "C" int iss_sidekit_tables(iss_ctx* c, const double* window400, const float* bank) {
    if (!c || !window400 || !bank) return iss_fail(c, ISS_EINVAL, "iss_sidekit_tables: NULL argument");
    ISS_HIP(c, hipSetDevice(c->device));
    // pack the sparse triangular bank: per filter {first_bin, n_bins, weight_offset}
    std::vector<int32_t> lim(24 * 3);
    std::vector<float> w;
    for (int f = 0; f < 24; ++f) {
        int lo = -1, hi = -1;
        for (int k = 0; k < 257; ++k)
            if (bank[f * 257 + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) { lo = 0; hi = -1; }
        if (hi >= 256) return iss_fail(c, ISS_EINVAL, "mel filter %d touches bin 256; unsupported table", f);
        lim[f * 3 + 0] = lo; lim[f * 3 + 1] = hi - lo + 1; lim[f * 3 + 2] = (int32_t)w.size();
        for (int k = lo; k <= hi; ++k) w.push_back(bank[f * 257 + k]);   // interior zeros kept (there are none)
    }
    if (w.size() > 1024) return iss_fail(c, ISS_EINVAL, "mel bank has %zu weights (>1024)", w.size());
    std::vector<double> tw(2 * 256 * 2);
    for (int k = 0; k < 256; ++k) {
        tw[2 * k] = cos(-2.0 * M_PI * k / 256.0);       tw[2 * k + 1] = sin(-2.0 * M_PI * k / 256.0);
        tw[512 + 2 * k] = cos(-2.0 * M_PI * k / 512.0); tw[512 + 2 * k + 1] = sin(-2.0 * M_PI * k / 512.0);
    }
    if (!c->d_window) ISS_HIP(c, hipMalloc((void**)&c->d_window, 400 * sizeof(double)));
    if (!c->d_melw) ISS_HIP(c, hipMalloc((void**)&c->d_melw, 1024 * sizeof(float)));
    if (!c->d_mellim) ISS_HIP(c, hipMalloc((void**)&c->d_mellim, 72 * sizeof(int32_t)));
    if (!c->d_tw) ISS_HIP(c, hipMalloc((void**)&c->d_tw, tw.size() * sizeof(double)));
    w.resize(1024, 0.0f);
    ISS_HIP(c, hipMemcpy(c->d_window, window400, 400 * sizeof(double), hipMemcpyHostToDevice));
    ISS_HIP(c, hipMemcpy(c->d_melw, w.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
    ISS_HIP(c, hipMemcpy(c->d_mellim, lim.data(), 72 * sizeof(int32_t), hipMemcpyHostToDevice));
    ISS_HIP(c, hipMemcpy(c->d_tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    c->sk_tables = true;
    return ISS_OK;
}

// ---------------------------------------------------------------- signal ingest
static int put_signal(iss_ctx* c, const void* host, int64_t n, size_t esz, int kind) {
    if (!c || (!host && n > 0) || n < 0) return iss_fail(c, ISS_EINVAL, "iss_signal: bad argument");
    ISS_HIP(c, hipSetDevice(c->device));
    int rc = iss_reserve(c, c->sig, (size_t)n * esz + 16);
    if (rc) return rc;
    if (n > 0) ISS_HIP(c, hipMemcpyAsync(c->sig.p, host, (size_t)n * esz, hipMemcpyHostToDevice, c->stream));
    c->sig_ptr = c->sig.p; c->sig_kind = kind; c->sig_n = n; c->have_feats = false;
    return ISS_OK;
}
extern "C" int iss_signal_pcm16(iss_ctx* c, const int16_t* pcm, int64_t n) { return put_signal(c, pcm, n, 2, 1); }
extern "C" int iss_signal_f32(iss_ctx* c, const float* sig, int64_t n) { return put_signal(c, sig, n, 4, 2); }
extern "C" int iss_signal_pcm16_device_stream(iss_ctx* c, const void* dev, int64_t n, void* producer_stream) {
    if (!c || (!dev && n > 0) || n < 0) return iss_fail(c, ISS_EINVAL, "iss_signal_pcm16_device: bad argument");
    ISS_HIP(c, hipSetDevice(c->device));
    if (n > 0) {
        hipPointerAttribute_t at;
        hipError_t e = hipPointerGetAttributes(&at, dev);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return iss_fail(c, ISS_EINVAL, "iss_signal_pcm16_device: %p is not a HIP allocation (%s)", dev, hipGetErrorString(e));
        }
        if (at.type != hipMemoryTypeDevice || at.device != c->device)
            return iss_fail(c, ISS_EINVAL, "iss_signal_pcm16_device: pointer belongs to %s %d, the context runs on device %d",
                            at.type == hipMemoryTypeDevice ? "device" : "host / managed memory of device", at.device, c->device);
    }
    // the library's stream is non-blocking: order it behind the producer explicitly (see iss.h)
    if (!c->order_ev) ISS_HIP(c, hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming));
    ISS_HIP(c, hipEventRecord(c->order_ev, (hipStream_t)producer_stream));
    ISS_HIP(c, hipStreamWaitEvent(c->stream, c->order_ev, 0));
    c->sig_ptr = dev; c->sig_kind = 1; c->sig_n = n; c->have_feats = false;
    return ISS_OK;
}
extern "C" int iss_signal_pcm16_device(iss_ctx* c, const void* dev, int64_t n) {
    return iss_signal_pcm16_device_stream(c, dev, n, nullptr);
}

// ---------------------------------------------------------------- pinned host memory
extern "C" int iss_host_alloc(iss_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return iss_fail(c, ISS_EINVAL, "iss_host_alloc: NULL argument");
    *out = nullptr;
    ISS_HIP(c, hipSetDevice(c->device));
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return iss_fail(c, ISS_ENOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    return ISS_OK;
}
extern "C" int iss_host_free(iss_ctx* c, void* p) {
    if (!c) return ISS_EINVAL;
    if (!p) return ISS_OK;
    ISS_HIP(c, hipSetDevice(c->device));
    ISS_HIP(c, hipStreamSynchronize(c->stream));     // nothing of ours may still read / write it
    ISS_HIP(c, hipHostFree(p));
    return ISS_OK;
}

// Copy `bytes` of caller memory into a pinned staging buffer of the context (the caller may reuse its memory as soon
// as we return; the async H2D copy then reads the staging buffer).  A buffer is reused once the event recorded by
// iss_stage_mark behind its consumer has completed.
int iss_stage_host(iss_ctx* c, const void* src, size_t bytes, void** pinned_out, int* slot_out) {
    int slot = -1;
    for (size_t i = 0; i < c->staging.size(); ++i) {
        auto& s = c->staging[i];
        if (s.busy && hipEventQuery(s.done) == hipSuccess) s.busy = false;
        if (!s.busy && s.cap >= bytes && slot < 0) slot = (int)i;
    }
    if (slot < 0) {
        for (size_t i = 0; i < c->staging.size() && slot < 0; ++i)        // grow an idle one rather than piling up buffers
            if (!c->staging[i].busy) slot = (int)i;
        if (slot < 0) { c->staging.emplace_back(); slot = (int)c->staging.size() - 1; }
        auto& s = c->staging[slot];
        if (s.p) { (void)hipHostFree(s.p); s.p = nullptr; s.cap = 0; }
        const size_t want = bytes + (bytes >> 2) + 4096;
        hipError_t e = hipHostMalloc(&s.p, want, hipHostMallocDefault);
        if (e != hipSuccess) { s.p = nullptr; return iss_fail(c, ISS_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        s.cap = want;
        if (!s.done) ISS_HIP(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    memcpy(c->staging[slot].p, src, bytes);
    *pinned_out = c->staging[slot].p;
    *slot_out = slot;
    return ISS_OK;
}
void iss_stage_mark(iss_ctx* c, int slot) {
    auto& s = c->staging[slot];
    (void)hipEventRecord(s.done, c->stream);
    s.busy = true;
}

extern "C" int iss_wait(iss_ctx* c, int64_t ticket) {
    if (!c) return ISS_EINVAL;
    ISS_HIP(c, hipSetDevice(c->device));
    const int64_t last = c->ticket_base + (int64_t)c->ticket_ev.size() - 1;
    if (ticket < 0 || ticket >= last) {
        ISS_HIP(c, hipStreamSynchronize(c->stream));
        for (auto e : c->ticket_ev) c->ev_pool.push_back(e);
        c->ticket_base += (int64_t)c->ticket_ev.size();
        c->ticket_ev.clear();
        iss_prof_collect(c);
        return ISS_OK;
    }
    if (ticket < c->ticket_base) return ISS_OK;       // already waited for
    ISS_HIP(c, hipEventSynchronize(c->ticket_ev[(size_t)(ticket - c->ticket_base)]));
    return ISS_OK;
}

extern "C" int iss_sidekit(iss_ctx* c, int32_t* T_out) {
    if (!c) return ISS_EINVAL;
    if (!c->sk_tables) return iss_fail(c, ISS_ESTATE, "iss_sidekit: call iss_sidekit_tables first");
    if (c->sig_kind == 0) return iss_fail(c, ISS_ESTATE, "iss_sidekit: no resident signal");
    ISS_HIP(c, hipSetDevice(c->device));
    int64_t n = c->sig_n;
    int64_t T = n >= 400 ? (n - 400) / 160 + 1 : 0;
    if (T > 0x7fffffffLL / 24) return iss_fail(c, ISS_EINVAL, "signal too long (%lld frames)", (long long)T);
    c->T = (int32_t)T;
    int rc = iss_reserve(c, c->mspec, (size_t)(T > 0 ? T : 1) * 24 * sizeof(float));
    if (rc) return rc;
    rc = iss_reserve(c, c->loge, (size_t)(T > 0 ? T : 1) * sizeof(float));
    if (rc) return rc;
    if (T > 0) { rc = iss_launch_sidekit(c); if (rc) return rc; }
    c->have_feats = true;
    if (T_out) *T_out = (int32_t)T;
    return ISS_OK;
}

extern "C" int iss_get_loge(iss_ctx* c, float* out) {
    if (!c || !out) return iss_fail(c, ISS_EINVAL, "iss_get_loge: NULL");
    if (!c->have_feats) return iss_fail(c, ISS_ESTATE, "iss_get_loge: no features resident");
    ISS_HIP(c, hipSetDevice(c->device));
    if (c->T > 0) ISS_HIP(c, hipMemcpyAsync(out, c->loge.p, (size_t)c->T * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    return ISS_OK;
}

extern "C" int iss_get_mspec(iss_ctx* c, float* out) {
    if (!c || !out) return iss_fail(c, ISS_EINVAL, "iss_get_mspec: NULL");
    if (!c->have_feats) return iss_fail(c, ISS_ESTATE, "iss_get_mspec: no features resident");
    ISS_HIP(c, hipSetDevice(c->device));
    if (c->T > 0) ISS_HIP(c, hipMemcpyAsync(out, c->mspec.p, (size_t)c->T * 24 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    return ISS_OK;
}

extern "C" int iss_set_mspec(iss_ctx* c, const float* mspec, int32_t T) {
    if (!c || !mspec || T <= 0) return iss_fail(c, ISS_EINVAL, "iss_set_mspec: bad argument");
    ISS_HIP(c, hipSetDevice(c->device));
    int rc = iss_reserve(c, c->mspec, (size_t)T * 24 * sizeof(float));
    if (rc) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->mspec.p, mspec, (size_t)T * 24 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));   // caller may free `mspec` right after
    c->T = T; c->have_feats = true;
    return ISS_OK;
}

// ---------------------------------------------------------------- profiling
static hipEvent_t get_event(iss_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void iss_prof_begin(iss_ctx* c, int kind, double flops) {
    if (!c->prof) return;
    iss_ctx::Pending p; p.a = get_event(c); p.b = get_event(c); p.kind = kind; p.sub = -1; p.row = -1; p.flops = flops;
    (void)hipEventRecord(p.a, c->stream);
    c->pending.push_back(p);
}
void iss_prof_tag(iss_ctx* c, int sub) {
    if (!c->prof || c->pending.empty() || sub < 0 || sub >= ISS_PROF_KINDS) return;
    c->pending.back().sub = sub;
}
void iss_prof_row(iss_ctx* c, int row) {
    if (!c->prof || c->pending.empty() || row < 0 || row >= ISS_PROF_ROWS) return;
    c->pending.back().row = row;
}
void iss_prof_inst(iss_ctx* c, const char* fmt, ...) {
    if (!c->prof || c->pending.empty()) return;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    c->pending.back().inst = buf;
}
void iss_prof_end(iss_ctx* c) {
    if (!c->prof || c->pending.empty()) return;
    (void)hipEventRecord(c->pending.back().b, c->stream);
}
void iss_prof_collect(iss_ctx* c) {      // stream must be idle
    for (auto& p : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->prof_ms[p.kind] += ms; c->prof_launch[p.kind] += 1; c->prof_flops[p.kind] += p.flops;
            if (p.sub >= 0) { c->prof_ms[p.sub] += ms; c->prof_launch[p.sub] += 1; c->prof_flops[p.sub] += p.flops; }
            if (p.row >= 0) { c->prof_row_ms[p.row] += ms; c->prof_row_launch[p.row] += 1; }
            if (!p.inst.empty()) {
                iss_ctx::Inst* e = nullptr;
                for (auto& i : c->prof_inst) if (i.name == p.inst) { e = &i; break; }
                if (!e) { c->prof_inst.emplace_back(); e = &c->prof_inst.back(); e->name = p.inst; }
                e->ms += ms; e->launches += 1; e->flops += p.flops;
            }
        }
        c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
}
extern "C" int iss_prof_enable(iss_ctx* c, int on) { if (!c) return ISS_EINVAL; c->prof = on != 0; return ISS_OK; }
extern "C" int iss_prof_reset(iss_ctx* c) {
    if (!c) return ISS_EINVAL;
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    for (int i = 0; i < ISS_PROF_KINDS; ++i) { c->prof_ms[i] = 0; c->prof_launch[i] = 0; c->prof_flops[i] = 0; }
    for (int i = 0; i < ISS_PROF_ROWS; ++i) { c->prof_row_ms[i] = 0; c->prof_row_launch[i] = 0; }
    c->prof_inst.clear();
    return ISS_OK;
}
extern "C" int iss_prof_get_instance(iss_ctx* c, int index, char* name_out, int32_t name_len, double* ms, int64_t* launches, double* flops) {
    if (!c || index < 0) return ISS_EINVAL;
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    if (index >= (int)c->prof_inst.size()) return ISS_EINVAL;          // (no error text: this is how a caller finds the end)
    const iss_ctx::Inst& e = c->prof_inst[(size_t)index];
    if (name_out && name_len > 0) { snprintf(name_out, (size_t)name_len, "%s", e.name.c_str()); }
    if (ms) *ms = e.ms;
    if (launches) *launches = e.launches;
    if (flops) *flops = e.flops;
    return ISS_OK;
}
extern "C" int iss_prof_get_row(iss_ctx* c, int row, double* ms, int64_t* launches) {
    if (!c || row < 0 || row >= ISS_PROF_ROWS) return ISS_EINVAL;
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    if (ms) *ms = c->prof_row_ms[row];
    if (launches) *launches = c->prof_row_launch[row];
    return ISS_OK;
}
extern "C" int iss_prof_get(iss_ctx* c, int kind, double* ms, int64_t* launches, double* flops) {
    if (!c || kind < 0 || kind >= ISS_PROF_KINDS) return ISS_EINVAL;
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    if (ms) *ms = c->prof_ms[kind];
    if (launches) *launches = c->prof_launch[kind];
    if (flops) *flops = c->prof_flops[kind];
    return ISS_OK;
}
