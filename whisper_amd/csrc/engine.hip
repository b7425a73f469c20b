// engine.hip — engine object + C-ABI (include/whisper_amd.h) for the MI355X local-Whisper path.
// Owns weights, workspace and KV caches in HBM; orchestrates mel -> encoder -> cross-K/V -> greedy decode as
// plain kernel launches on the caller's HIP stream, with the decode step captured once into a hipGraph.
#include "common.h"
#include "../../include/whisper_amd.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace wx;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(WX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define WX_TRY(expr)                                                                                    \
    do {                                                                                                \
        int _r = (expr);                                                                                \
        if (_r != 0) return fail(_r == -2 ? WX_ERR_HIP : WX_ERR_INVALID, std::string(#expr) + " failed (" + std::to_string(_r) + ")"); \
    } while (0)

namespace {

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

struct AttnW { void* wqkv = nullptr; float* bqkv = nullptr; void* wo = nullptr; float* bo = nullptr;
               void *wqkv_pk = nullptr, *wo_pk = nullptr; };   // _pk: decoder only, MFMA-fragment order (GemmArgs::Wpk)
struct LayerW {
    float *ln1_g = nullptr, *ln1_b = nullptr;
    AttnW self;
    float *lnx_g = nullptr, *lnx_b = nullptr;      // cross-attention LN (decoder only)
    void* xq = nullptr; float* xbq = nullptr;      // cross query
    void* xkv = nullptr; float* xbkv = nullptr;    // fused cross key|value
    void* xk_pk = nullptr;                         // bf16 engines: cross key weight, transposed, per head in MFMA-fragment order (xattn.hip)
    void* xv_pk = nullptr;                         // bf16 engines: cross value weight in MFMA-fragment order (xattn.hip)
    void* xo = nullptr; float* xbo = nullptr;
    float *ln2_g = nullptr, *ln2_b = nullptr;
    void* w1 = nullptr; float* b1 = nullptr; void* w2 = nullptr; float* b2 = nullptr;
    void *w1_pk = nullptr, *w2_pk = nullptr, *xq_pk = nullptr, *xo_pk = nullptr;
};

// Device allocations that hold WEIGHTS (and the read-only mel tables): owned by a block that every engine built from the same
// weights refers to (wx_create_shared), freed with the last of them.
struct WeightBlock {
    int device = 0;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    ~WeightBlock() {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct wx_engine {
    wx_config cfg{};
    int dt = DT_F32;
    bool finalized = false;
    std::map<std::string, HostTensor> staged;
    std::vector<float> mel_filters_h;
    std::vector<int32_t> suppress_h;
    std::vector<void*> allocs;                               // workspace, caches, decode state: this engine's own
    std::shared_ptr<WeightBlock> wb;                         // weights: shared with the engines wx_create_shared made from this one
    bool owns_weights = true;                                // false: built by wx_create_shared (wx_weight_bytes reports 0)
    int64_t weight_bytes = 0, work_bytes = 0;

    // dims
    int d = 0, H = 0, dt_ = 0, Ht = 0, c_pad = 0, vocab_pad = 0, T = 0, n_pages_seq = 0;
    int ld_tok = 0;     // ids per token row = n_text_ctx + 1: upstream's loop still samples from position n_text_ctx - 1
    float qk_scale = 1.f;

    // weights
    void *conv1_w = nullptr, *conv2_w = nullptr; float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr;
    std::vector<LayerW> enc, dec;
    float *lnpost_g = nullptr, *lnpost_b = nullptr, *declnf_g = nullptr, *declnf_b = nullptr;
    void *tok_emb = nullptr, *dec_pos = nullptr, *tok_emb_pk = nullptr;   // _pk: MFMA-fragment order (GemmArgs::Wpk)
    MelTables mel{};
    uint8_t* suppress = nullptr;

    // encoder workspace
    void *xpad1 = nullptr, *xpad2 = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *enc_out = nullptr;
    float* x = nullptr;
    // decoder state
    void *crossK = nullptr, *crossV = nullptr, *selfK = nullptr, *selfV = nullptr;
    int32_t* page_table = nullptr;
    float* dx = nullptr; void *dxn = nullptr, *dqkv = nullptr, *datt = nullptr, *dhid = nullptr, *dq = nullptr;
    float* logits = nullptr;
    int32_t *tokens = nullptr, *d_pos = nullptr, *d_counter = nullptr, *n_finished = nullptr, *n_tokens = nullptr;
    int32_t *row_n_prompt = nullptr, *row_sot = nullptr;      // per-row prompt length / SOT position
    uint32_t* dyn = nullptr;                                 // {temperature bits, seed, sample_len} read by the sampler
    float *sum_logprob = nullptr, *no_speech = nullptr, *partial = nullptr, *gpartial = nullptr, *samp_part = nullptr;
    // decoder cross-attention on the encoder output (xattn.hip; bf16 engines, every row whose window has no beams to share with)
    bool xattn = false;
    int pass_rows = 0;                                       // rows per decode pass of the split-K path: max_batch, or its even share of
                                                             // ceil(max_batch / DG_MAX_ROWS) passes rounded up to 16 (a function of max_batch only)
    int x_split = 1;                                         // key splits per row: a function of max_batch only (batch invariance)
    void* enc_frag = nullptr;                                // encoder output in MFMA-fragment order [windows][x_nt16][d / 32][64][8] bf16
    int x_nt16 = 0;                                          // 16-key tiles per window in enc_frag (even; zero rows past T)
    void *x_qt = nullptr, *x_cpart = nullptr; float* x_ml = nullptr;
    int cross_kv_B = 0;                                      // windows whose K/V cache matches the last wx_encode (lazy: beams, > 80 rows)
    int32_t* h_pinned = nullptr;
    float* static_scratch = nullptr;                         // this device's address of the file-scope `scratch` array (wx_log_mel_static)
    float* rs_taps = nullptr;                                // resampler taps (wx_resample_pcm16), RS_MAX_TAPS floats
    int encoded_B = 0;
    // beam search state (rows = windows x beams <= max_batch)
    int beam_n = 0;                      // > 0 while a beam decode is building / replaying its step
    int32_t *b_tokens_tmp = nullptr, *b_node = nullptr, *b_node_tmp = nullptr, *b_cand_tok = nullptr, *b_src = nullptr, *b_next = nullptr;
    int32_t *b_fin_tokens = nullptr, *b_fin_count = nullptr;
    float *b_sum_tmp = nullptr, *b_cand_score = nullptr, *b_fin_score = nullptr;

    hipStream_t own_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // cached decode-step graph
    hipGraphExec_t graph_exec = nullptr;
    std::vector<int> graph_key;
    // optional event brackets around cross-attention launches (eager mode)
    std::vector<hipEvent_t> prof_events;
    size_t prof_used = 0;
    bool prof_on = false;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
};

namespace {

int dev_alloc(wx_engine* e, void** p, size_t bytes, bool weight) {
    if (bytes == 0) bytes = 16;
    hipError_t r = hipMalloc(p, bytes);
    if (r != hipSuccess) return fail(WX_ERR_NOMEM, "hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(r));
    if (weight) { e->wb->allocs.push_back(*p); e->wb->bytes += (int64_t)bytes; e->weight_bytes += (int64_t)bytes; }
    else { e->allocs.push_back(*p); e->work_bytes += (int64_t)bytes; }
    return 0;
}

// upload fp32 host data as fp32
int upload_f32(wx_engine* e, const float* h, size_t n, float** out) {
    void* p;
    int r = dev_alloc(e, &p, n * 4, true);
    if (r) return r;
    HIP_TRY(hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice));
    *out = (float*)p;
    return 0;
}
// upload fp32 host data converted to the engine dtype
int upload_w(wx_engine* e, const float* h, size_t n, void** out) {
    if (e->dt == DT_F32) return upload_f32(e, h, n, (float**)out);
    void* tmp;
    HIP_TRY(hipMalloc(&tmp, n * 4));
    hipError_t r1 = hipMemcpy(tmp, h, n * 4, hipMemcpyHostToDevice);
    void* p = nullptr;
    int r = dev_alloc(e, &p, n * 2, true);
    if (r == 0 && r1 == hipSuccess) r = launch_convert_f32((const float*)tmp, p, DT_BF16, (int64_t)n, nullptr);
    hipError_t r2 = hipDeviceSynchronize();
    (void)hipFree(tmp);
    if (r1 != hipSuccess || r2 != hipSuccess) return fail(WX_ERR_HIP, "weight upload failed");
    if (r) return r;
    *out = p;
    return 0;
}

// second copy of a decode-GEMM weight in the order its MFMA fragments are loaded (bf16 engines; no-op otherwise)
int pack_w(wx_engine* e, const void* w, int N, int K, void** out) {
    static const bool no_pack = getenv("WX_NO_PACK") != nullptr;
    *out = nullptr;
    if (e->dt != DT_BF16 || no_pack || N % 16 || K % 32) return 0;
    void* p = nullptr;
    int r = dev_alloc(e, &p, (size_t)N * K * 2, true);
    if (r) return r;
    if (launch_pack_gemm_weight(w, K, N, K, p, nullptr)) return fail(WX_ERR_INVALID, "weight repack refused");
    HIP_TRY(hipDeviceSynchronize());
    *out = p;
    return 0;
}

const HostTensor* get(wx_engine* e, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = e->staged.find(name);
    if (it == e->staged.end()) { fail(WX_ERR_STATE, "missing weight tensor " + name); return nullptr; }
    const HostTensor& t = it->second;
    if (t.shape != std::vector<int64_t>(shape)) { fail(WX_ERR_INVALID, "wrong shape for " + name); return nullptr; }
    return &t;
}

int build_attn(wx_engine* e, const std::string& p, int w, AttnW& out) {
    const HostTensor *q = get(e, p + ".query.weight", {w, w}), *qb = get(e, p + ".query.bias", {w});
    const HostTensor *k = get(e, p + ".key.weight", {w, w});
    const HostTensor *v = get(e, p + ".value.weight", {w, w}), *vb = get(e, p + ".value.bias", {w});
    const HostTensor *o = get(e, p + ".out.weight", {w, w}), *ob = get(e, p + ".out.bias", {w});
    if (!q || !qb || !k || !v || !vb || !o || !ob) return WX_ERR_STATE;
    std::vector<float> fw((size_t)3 * w * w), fb((size_t)3 * w, 0.f);
    memcpy(fw.data(), q->data.data(), (size_t)w * w * 4);
    memcpy(fw.data() + (size_t)w * w, k->data.data(), (size_t)w * w * 4);
    memcpy(fw.data() + (size_t)2 * w * w, v->data.data(), (size_t)w * w * 4);
    memcpy(fb.data(), qb->data.data(), (size_t)w * 4);
    memcpy(fb.data() + 2 * w, vb->data.data(), (size_t)w * 4);
    int r;
    if ((r = upload_w(e, fw.data(), fw.size(), &out.wqkv))) return r;
    if ((r = upload_f32(e, fb.data(), fb.size(), &out.bqkv))) return r;
    if ((r = upload_w(e, o->data.data(), o->data.size(), &out.wo))) return r;
    if ((r = upload_f32(e, ob->data.data(), ob->data.size(), &out.bo))) return r;
    return 0;
}

int build_ln(wx_engine* e, const std::string& p, int w, float** g, float** b) {
    const HostTensor *gw = get(e, p + ".weight", {w}), *gb = get(e, p + ".bias", {w});
    if (!gw || !gb) return WX_ERR_STATE;
    int r;
    if ((r = upload_f32(e, gw->data.data(), w, g))) return r;
    return upload_f32(e, gb->data.data(), w, b);
}

int build_mlp(wx_engine* e, const std::string& p, int w, LayerW& L) {
    const HostTensor *w1 = get(e, p + ".mlp.0.weight", {4 * w, w}), *b1 = get(e, p + ".mlp.0.bias", {4 * w});
    const HostTensor *w2 = get(e, p + ".mlp.2.weight", {w, 4 * w}), *b2 = get(e, p + ".mlp.2.bias", {w});
    if (!w1 || !b1 || !w2 || !b2) return WX_ERR_STATE;
    int r;
    if ((r = upload_w(e, w1->data.data(), w1->data.size(), &L.w1))) return r;
    if ((r = upload_f32(e, b1->data.data(), b1->data.size(), &L.b1))) return r;
    if ((r = upload_w(e, w2->data.data(), w2->data.size(), &L.w2))) return r;
    return upload_f32(e, b2->data.data(), b2->data.size(), &L.b2);
}

// conv weight [out][in][3] -> [out][3][c_pad] (tap-major, channel-last, zero padded channels)
int build_conv(wx_engine* e, const std::string& p, int out_c, int in_c, int c_pad, void** w, float** b) {
    const HostTensor *cw = get(e, p + ".weight", {out_c, in_c, 3}), *cb = get(e, p + ".bias", {out_c});
    if (!cw || !cb) return WX_ERR_STATE;
    std::vector<float> r((size_t)out_c * 3 * c_pad, 0.f);
    for (int o = 0; o < out_c; ++o)
        for (int c = 0; c < in_c; ++c)
            for (int k = 0; k < 3; ++k) r[((size_t)o * 3 + k) * c_pad + c] = cw->data[((size_t)o * in_c + c) * 3 + k];
    int rc;
    if ((rc = upload_w(e, r.data(), r.size(), w))) return rc;
    return upload_f32(e, cb->data.data(), out_c, b);
}

GemmArgs gemm_basic(wx_engine* e, const void* A, int64_t lda, const void* W, int M, int N, int K) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.in_dtype = e->dt;
    return g;
}

// Every hot entry point runs with the ENGINE's device current and restores the caller's on the way out: a process that
// drives several GPUs (or never called hipSetDevice / torch.cuda.set_device for this one) must not launch on, or allocate
// from, the wrong device.
struct DeviceScope {
    int prev = -1, dev;
    explicit DeviceScope(const wx_engine* e) : dev(e->cfg.device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

struct StreamScope {   // graph capture needs a real stream: route NULL-stream callers through the engine's own stream
    wx_engine* e; hipStream_t user, st; bool swapped;
    StreamScope(wx_engine* e_, void* s) : e(e_), user((hipStream_t)s), st((hipStream_t)s), swapped(false) {
        if (user == nullptr) {
            st = e->own_stream; swapped = true;
            (void)hipEventRecord(e->ev_in, user);
            (void)hipStreamWaitEvent(st, e->ev_in, 0);
        }
    }
    ~StreamScope() {
        if (swapped) { (void)hipEventRecord(e->ev_out, st); (void)hipStreamWaitEvent(user, e->ev_out, 0); }
    }
};

}  // namespace

// ---- the allocation-free entry point (include/whisper_amd.h: wx_log_mel_static) ------------------------------------------
// Its only device memory beyond input and output is this file-scope array: slot 0 holds the buffer's maximum, slots
// WX_STATIC_HEAD ... the per-workgroup maxima of the stage-1 launch.  Every GPU the library runs on holds its own copy at
// its own address until the process ends (256 KiB).  `scratch` itself is a host-side stand-in for the symbol, NOT a device
// address: host code only ever uses the per-device pointer that hipGetSymbolAddress returned (resolved in wx_create with
// the engine's device current and kept in the engine).
constexpr int WX_STATIC_FLOATS = 65536, WX_STATIC_HEAD = 64;
static __device__ __align__(256) float scratch[WX_STATIC_FLOATS];

static int resolve_static_scratch(wx_engine* e) {       // caller holds a DeviceScope on e
    float* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(WX_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    e->static_scratch = ptr;
    return 0;
}

extern "C" {

const char* wx_last_error(void) { return g_err.c_str(); }
const char* wx_version(void) { return "whisper_amd 0.1 (gfx950)"; }

int wx_create(const wx_config* cfg, wx_engine** out) {
    if (!cfg || !out) return fail(WX_ERR_INVALID, "null argument");
    const wx_config& c = *cfg;
    if (c.n_audio_state % 128 || c.n_text_state % 128 || c.n_audio_state != c.n_audio_head * HEAD_DIM ||
        c.n_text_state != c.n_text_head * HEAD_DIM || c.n_audio_state != c.n_text_state)
        return fail(WX_ERR_INVALID, "state widths must be equal, multiples of 128, with 64-wide heads");
    if (c.n_audio_ctx != 1500) return fail(WX_ERR_INVALID, "n_audio_ctx must be 1500 (30 s windows)");
    if (c.n_text_ctx % KV_PAGE || c.n_text_ctx <= 0 || c.n_mels <= 0 || c.n_mels > 256 || c.max_batch <= 0 || c.n_vocab <= 0)
        return fail(WX_ERR_INVALID, "bad dimensions");
    if (c.precision != WX_PREC_F32 && c.precision != WX_PREC_BF16) return fail(WX_ERR_INVALID, "bad precision");
    {   // the device must exist; streams and events below are created on it
        int n_dev = 0;
        HIP_TRY(hipGetDeviceCount(&n_dev));
        if (c.device < 0 || c.device >= n_dev) return fail(WX_ERR_INVALID, "no such device");
    }
    wx_engine* e = new wx_engine();
    e->cfg = c;
    e->wb = std::make_shared<WeightBlock>();
    e->wb->device = c.device;
    DeviceScope dev_scope(e);
    e->dt = c.precision == WX_PREC_BF16 ? DT_BF16 : DT_F32;
    e->d = c.n_audio_state; e->H = c.n_audio_head; e->dt_ = c.n_text_state; e->Ht = c.n_text_head;
    e->c_pad = (c.n_mels + 63) / 64 * 64;
    e->vocab_pad = (c.n_vocab + 15) / 16 * 16;
    e->T = c.n_audio_ctx;
    e->n_pages_seq = c.n_text_ctx / KV_PAGE;
    e->ld_tok = c.n_text_ctx + 1;
    e->qk_scale = powf((float)HEAD_DIM, -0.25f);
    // Cross-attention on the encoder output needs enough rows to fill the chip with one workgroup per (row, key split <= 8): from
    // 32 windows up it wins (1257 vs 1254 RTFx at 32, 1544 vs 1477 at 48, 1801 vs 1572 at 64, 1937 vs 1643 at 80), below it the K/V
    // form's (window, head, split) grid does (853 vs 739 at 16).  An engine property, not a per-call choice: a window decodes to
    // the same ids whatever the batch it is in.  WX_XATTN=1 / WX_NO_XATTN=1 force either form (tests, A/B).
    e->xattn = e->dt == DT_BF16 && getenv("WX_NO_XATTN") == nullptr && (c.max_batch >= 32 || getenv("WX_XATTN") != nullptr);
    hipError_t r = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (r == hipSuccess) r = hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming);
    if (r == hipSuccess) r = hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming);
    if (r != hipSuccess) { delete e; return fail(WX_ERR_HIP, hipGetErrorString(r)); }
    if (int rs = resolve_static_scratch(e)) { wx_destroy(e); return rs; }     // once per engine, with its device current
    *out = e;
    return WX_OK;
}

void wx_destroy(wx_engine* e) {
    if (!e) return;
    DeviceScope dev_scope(e);
    (void)hipDeviceSynchronize();
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    for (hipEvent_t ev : e->prof_events) (void)hipEventDestroy(ev);
    for (void* p : e->allocs) (void)hipFree(p);
    e->wb.reset();                                           // frees the weights with the last engine that uses them
    if (e->h_pinned) (void)hipHostFree(e->h_pinned);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    delete e;
}

int wx_set_weight(wx_engine* e, const char* name, const float* h_data, const int64_t* shape, int32_t ndim) {
    if (!e || !name || !h_data || !shape || ndim <= 0 || ndim > 4) return fail(WX_ERR_INVALID, "bad argument");
    if (e->finalized) return fail(WX_ERR_STATE, "engine already finalized");
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] <= 0) return fail(WX_ERR_INVALID, "bad shape"); t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(h_data, h_data + n);
    e->staged[name] = std::move(t);
    return WX_OK;
}

int wx_set_mel_filters(wx_engine* e, const float* h_filters) {
    if (!e || !h_filters) return fail(WX_ERR_INVALID, "bad argument");
    if (e->finalized) return fail(WX_ERR_STATE, "engine already finalized");
    e->mel_filters_h.assign(h_filters, h_filters + (size_t)e->cfg.n_mels * N_FREQ);
    return WX_OK;
}

int wx_set_suppress_tokens(wx_engine* e, const int32_t* h_ids, int32_t n) {
    if (!e || (n > 0 && !h_ids) || n < 0) return fail(WX_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) if (h_ids[i] < 0 || h_ids[i] >= e->cfg.n_vocab) return fail(WX_ERR_INVALID, "suppress id out of range");
    e->suppress_h.assign(h_ids, h_ids + n);
    if (e->finalized) {
        // the mask is read by every sampler launch, the captured step graph included: wait for whatever the engine's
        // device still has in flight before it changes (the caller must not change it WHILE a decode call is running)
        DeviceScope dev_scope(e);
        std::vector<uint8_t> m(e->vocab_pad, 0);
        for (int id : e->suppress_h) m[id] = 1;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(e->suppress, m.data(), m.size(), hipMemcpyHostToDevice));
    }
    return WX_OK;
}

}  // extern "C"

// weights and read-only tables -> the engine's weight block (caller holds a DeviceScope)
static int build_weights(wx_engine* e) {
    const wx_config& c = e->cfg;
    const int d = e->d, T = e->T;
    int r;
    if ((r = build_conv(e, "encoder.conv1", d, c.n_mels, e->c_pad, &e->conv1_w, &e->conv1_b))) return r;
    if ((r = build_conv(e, "encoder.conv2", d, d, d, &e->conv2_w, &e->conv2_b))) return r;
    {
        const HostTensor* pe = get(e, "encoder.positional_embedding", {T, d});
        if (!pe) return WX_ERR_STATE;
        if ((r = upload_f32(e, pe->data.data(), pe->data.size(), &e->enc_pos))) return r;
    }
    e->enc.resize(c.n_audio_layer);
    for (int i = 0; i < c.n_audio_layer; ++i) {
        const std::string p = "encoder.blocks." + std::to_string(i);
        LayerW& L = e->enc[i];
        if ((r = build_ln(e, p + ".attn_ln", d, &L.ln1_g, &L.ln1_b))) return r;
        if ((r = build_attn(e, p + ".attn", d, L.self))) return r;
        if ((r = build_ln(e, p + ".mlp_ln", d, &L.ln2_g, &L.ln2_b))) return r;
        if ((r = build_mlp(e, p, d, L))) return r;
    }
    if ((r = build_ln(e, "encoder.ln_post", d, &e->lnpost_g, &e->lnpost_b))) return r;
    e->dec.resize(c.n_text_layer);
    for (int i = 0; i < c.n_text_layer; ++i) {
        const std::string p = "decoder.blocks." + std::to_string(i);
        LayerW& L = e->dec[i];
        if ((r = build_ln(e, p + ".attn_ln", d, &L.ln1_g, &L.ln1_b))) return r;
        if ((r = build_attn(e, p + ".attn", d, L.self))) return r;
        if ((r = build_ln(e, p + ".cross_attn_ln", d, &L.lnx_g, &L.lnx_b))) return r;
        {
            const std::string x = p + ".cross_attn";
            const HostTensor *q = get(e, x + ".query.weight", {d, d}), *qb = get(e, x + ".query.bias", {d});
            const HostTensor *k = get(e, x + ".key.weight", {d, d});
            const HostTensor *v = get(e, x + ".value.weight", {d, d}), *vb = get(e, x + ".value.bias", {d});
            const HostTensor *o = get(e, x + ".out.weight", {d, d}), *ob = get(e, x + ".out.bias", {d});
            if (!q || !qb || !k || !v || !vb || !o || !ob) return WX_ERR_STATE;
            std::vector<float> kv((size_t)2 * d * d), kvb((size_t)2 * d, 0.f);
            memcpy(kv.data(), k->data.data(), (size_t)d * d * 4);
            memcpy(kv.data() + (size_t)d * d, v->data.data(), (size_t)d * d * 4);
            memcpy(kvb.data() + d, vb->data.data(), (size_t)d * 4);
            if ((r = upload_w(e, q->data.data(), q->data.size(), &L.xq))) return r;
            if ((r = upload_f32(e, qb->data.data(), d, &L.xbq))) return r;
            if ((r = upload_w(e, kv.data(), kv.size(), &L.xkv))) return r;
            if (e->xattn) {
                // fragment order of v_mfma_f32_16x16x32_bf16's A operand: lane (g, r) of a block holds row r, k = 8g .. 8g + 7
                //   key:   [head][d / 16 tiles of E's feature index i][2 k-steps over the head's 64 outputs c]: Wk[64 h + 32 ks + 8 g + j][16 it + r]
                //   value: [d / 16 tiles of the output index n][d / 32 k-steps over E's feature index]:        Wv[16 u + r][32 kb + 8 g + j]
                std::vector<float> kp((size_t)d * d), vp((size_t)d * d);
                size_t at = 0;
                for (int h = 0; h < d / 64; ++h)
                    for (int it = 0; it < d / 16; ++it)
                        for (int ks = 0; ks < 2; ++ks)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j)
                                    kp[at++] = k->data[(size_t)(h * 64 + ks * 32 + (lane >> 4) * 8 + j) * d + it * 16 + (lane & 15)];
                at = 0;
                for (int u = 0; u < d / 16; ++u)
                    for (int kb = 0; kb < d / 32; ++kb)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j)
                                vp[at++] = v->data[(size_t)(u * 16 + (lane & 15)) * d + kb * 32 + (lane >> 4) * 8 + j];
                if ((r = upload_w(e, kp.data(), kp.size(), &L.xk_pk)) || (r = upload_w(e, vp.data(), vp.size(), &L.xv_pk))) return r;
            }
            if ((r = upload_f32(e, kvb.data(), kvb.size(), &L.xbkv))) return r;
            if ((r = upload_w(e, o->data.data(), o->data.size(), &L.xo))) return r;
            if ((r = upload_f32(e, ob->data.data(), d, &L.xbo))) return r;
        }
        if ((r = build_ln(e, p + ".mlp_ln", d, &L.ln2_g, &L.ln2_b))) return r;
        if ((r = build_mlp(e, p, d, L))) return r;
        if ((r = pack_w(e, L.self.wqkv, 3 * d, d, &L.self.wqkv_pk)) || (r = pack_w(e, L.self.wo, d, d, &L.self.wo_pk)) ||
            (r = pack_w(e, L.xq, d, d, &L.xq_pk)) || (r = pack_w(e, L.xo, d, d, &L.xo_pk)) ||
            (r = pack_w(e, L.w1, 4 * d, d, &L.w1_pk)) || (r = pack_w(e, L.w2, d, 4 * d, &L.w2_pk))) return r;
    }
    if ((r = build_ln(e, "decoder.ln", d, &e->declnf_g, &e->declnf_b))) return r;
    {
        const HostTensor* te = get(e, "decoder.token_embedding.weight", {c.n_vocab, d});
        const HostTensor* pe = get(e, "decoder.positional_embedding", {c.n_text_ctx, d});
        if (!te || !pe) return WX_ERR_STATE;
        std::vector<float> padded((size_t)e->vocab_pad * d, 0.f);
        memcpy(padded.data(), te->data.data(), te->data.size() * 4);
        if ((r = upload_w(e, padded.data(), padded.size(), &e->tok_emb))) return r;
        if ((r = pack_w(e, e->tok_emb, e->vocab_pad, d, &e->tok_emb_pk))) return r;   // for the vocabulary projection
        if ((r = upload_w(e, pe->data.data(), pe->data.size(), &e->dec_pos))) return r;
    }
    e->staged.clear();
    // ---- mel tables ----
    {
        std::vector<float> win(N_FFT), cs(N_FFT * 2);
        for (int i = 0; i < N_FFT; ++i) {
            win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / N_FFT));
            cs[2 * i] = (float)cos(2.0 * M_PI * i / N_FFT);
            cs[2 * i + 1] = (float)sin(2.0 * M_PI * i / N_FFT);
        }
        // sparse filterbank: per mel bin {first non-zero DFT bin, last + 1, offset of its weights in the packed table, 0}
        std::vector<int32_t> fr((size_t)c.n_mels * 4, 0);
        std::vector<float> pack;
        for (int m = 0; m < c.n_mels; ++m) {
            int lo = N_FREQ, hi = 0;
            for (int k = 0; k < N_FREQ; ++k) if (e->mel_filters_h[(size_t)m * N_FREQ + k] != 0.f) { if (k < lo) lo = k; hi = k + 1; }
            if (lo > hi) lo = hi = 0;
            fr[4 * m] = lo; fr[4 * m + 1] = hi; fr[4 * m + 2] = (int32_t)pack.size();
            for (int k = lo; k < hi; ++k) pack.push_back(e->mel_filters_h[(size_t)m * N_FREQ + k]);
        }
        if (pack.empty()) pack.push_back(0.f);
        float *w, *t, *f; void* fp;
        if ((r = upload_f32(e, win.data(), win.size(), &w))) return r;
        if ((r = upload_f32(e, cs.data(), cs.size(), &t))) return r;
        if ((r = upload_f32(e, pack.data(), pack.size(), &f))) return r;
        if ((r = dev_alloc(e, &fp, fr.size() * 4, true))) return r;
        HIP_TRY(hipMemcpy(fp, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
        e->mel.n_pack = (int)pack.size();
        e->mel.window = w; e->mel.cs = t; e->mel.filters = f; e->mel.frange = (const int32_t*)fp; e->mel.n_mels = c.n_mels;
    }
    return 0;
}

// workspace, caches and decode state of ONE engine (its own even when the weights are shared; caller holds a DeviceScope)
static int build_workspace(wx_engine* e) {
    const wx_config& c = e->cfg;
    const int d = e->d, B = c.max_batch, T = e->T;
    const size_t es = dtype_size(e->dt);
    int r;
    {   // the suppress mask can be replaced per engine after wx_finalize (wx_set_suppress_tokens): never shared
        std::vector<uint8_t> m(e->vocab_pad, 0);       // padded: the sampler reads the mask 4 bytes at a time
        for (int id : e->suppress_h) m[id] = 1;
        void* p;
        if ((r = dev_alloc(e, &p, m.size(), false))) return r;
        HIP_TRY(hipMemcpy(p, m.data(), m.size(), hipMemcpyHostToDevice));
        e->suppress = (uint8_t*)p;
    }
    {
        const int passes = (B + DG_MAX_ROWS - 1) / DG_MAX_ROWS;
        e->pass_rows = passes <= 1 ? B : (((B + passes - 1) / passes + 15) / 16) * 16;
    }
    const int64_t rows = (int64_t)B * T;
    auto A = [&](void** p, size_t bytes) { return dev_alloc(e, p, bytes, false); };
    if ((r = A(&e->xpad1, (size_t)B * (WIN_FRAMES + 2) * e->c_pad * es))) return r;
    if ((r = A(&e->xpad2, (size_t)B * (WIN_FRAMES + 2) * d * es))) return r;
    HIP_TRY(hipMemset(e->xpad1, 0, (size_t)B * (WIN_FRAMES + 2) * e->c_pad * es));
    HIP_TRY(hipMemset(e->xpad2, 0, (size_t)B * (WIN_FRAMES + 2) * d * es));
    if ((r = A((void**)&e->x, (size_t)rows * d * 4))) return r;
    if ((r = A(&e->xn, (size_t)rows * d * es))) return r;
    if ((r = A(&e->qkv, (size_t)3 * rows * d * es))) return r;
    if ((r = A(&e->att, (size_t)rows * d * es))) return r;
    if ((r = A(&e->hid, (size_t)rows * 4 * d * es))) return r;
    if ((r = A(&e->enc_out, (size_t)rows * d * es))) return r;
    if (!e->xattn) {       // xattn engines attend the encoder output itself; their K/V cache is built on first need (ensure_cross_kv)
        const size_t cross_layer = (size_t)B * e->Ht * T * HEAD_DIM;
        if ((r = A(&e->crossK, cross_layer * c.n_text_layer * es))) return r;
        if ((r = A(&e->crossV, cross_layer * c.n_text_layer * es))) return r;
    } else {
        // key splits per row: as many as fit ONE round of workgroups (the stream kernel owns a CU: 256 workgroups), at most 8 (or
        // WX_XATTN_SPLIT); a function of max_batch only
        // (with n_lanes engines sharing the GPU, one round of THEIR workgroups together: large-v3 x 64 windows, 4 lanes: 2447 / 2552 /
        // 2652 / 2723 RTFx at 4 / 3 / 2 / 1 splits -- the stream kernel owns its CU's registers, so a launch that fills the chip
        // stalls every other lane's short launches for its whole duration)
        {
            // alone on the GPU: rounded DOWN, one full round of workgroups (48 rows: 5 splits = 240 workgroups, not 6 = 288 in two
            // rounds); sharing it: rounded up (3 lanes x 64 rows: 2 splits 2587 RTFx, 1 split 2473)
            const int share = (c.n_lanes > 1 ? c.n_lanes : 1) * e->pass_rows;
            e->x_split = c.n_lanes > 1 ? (256 + share - 1) / share : 256 / share;
        }
        if (e->x_split < 1) e->x_split = 1; if (e->x_split > 8) e->x_split = 8;
        if (const char* xs = getenv("WX_XATTN_SPLIT")) { const int v = atoi(xs); if (v >= 1 && v <= 8) e->x_split = v; }
        e->x_nt16 = ((T + 15) / 16 + 1) / 2 * 2;
        if ((r = A(&e->enc_frag, (size_t)B * e->x_nt16 * 16 * d * 2))) return r;
        if ((r = A(&e->x_qt, (size_t)B * e->Ht * d * 2))) return r;
        if ((r = A(&e->x_cpart, (size_t)B * e->x_split * e->Ht * d * 2))) return r;
        if ((r = A((void**)&e->x_ml, (size_t)B * e->x_split * e->Ht * 2 * 4))) return r;
    }
    const size_t self_layer = (size_t)B * e->n_pages_seq * e->Ht * KV_PAGE * HEAD_DIM;
    if ((r = A(&e->selfK, self_layer * c.n_text_layer * es))) return r;
    if ((r = A(&e->selfV, self_layer * c.n_text_layer * es))) return r;
    if ((r = A((void**)&e->page_table, (size_t)B * e->n_pages_seq * 4))) return r;
    {
        std::vector<int32_t> pt((size_t)B * e->n_pages_seq);
        for (size_t i = 0; i < pt.size(); ++i) pt[i] = (int32_t)i;    // identity allocation: sequence b owns pages b*P .. b*P+P-1
        HIP_TRY(hipMemcpy(e->page_table, pt.data(), pt.size() * 4, hipMemcpyHostToDevice));
    }
    if ((r = A((void**)&e->dx, (size_t)B * d * 4))) return r;
    if ((r = A(&e->dxn, (size_t)B * d * es))) return r;
    if ((r = A(&e->dqkv, (size_t)B * 3 * d * es))) return r;
    if ((r = A(&e->datt, (size_t)B * d * es))) return r;
    if ((r = A(&e->dhid, (size_t)B * 4 * d * es))) return r;
    if ((r = A(&e->dq, (size_t)B * d * es))) return r;
    if ((r = A((void**)&e->logits, (size_t)B * e->vocab_pad * 4))) return r;
    if ((r = A((void**)&e->tokens, (size_t)B * e->ld_tok * 4))) return r;
    if ((r = A((void**)&e->d_pos, 64))) return r;
    e->d_counter = e->d_pos + 4; e->n_finished = e->d_pos + 8;
    HIP_TRY(hipMemset(e->d_pos, 0, 64));
    if ((r = A((void**)&e->n_tokens, (size_t)B * 4))) return r;
    if ((r = A((void**)&e->row_n_prompt, (size_t)B * 4))) return r;
    if ((r = A((void**)&e->row_sot, (size_t)B * 4))) return r;
    if ((r = A((void**)&e->dyn, 64))) return r;
    if ((r = A((void**)&e->sum_logprob, (size_t)B * 4))) return r;
    if ((r = A((void**)&e->no_speech, (size_t)B * 4))) return r;
    if ((r = A((void**)&e->partial, (size_t)B * e->Ht * 8 * (HEAD_DIM + 2) * 4))) return r;
    if ((r = A((void**)&e->gpartial, (size_t)16 * B * 3 * d * 4))) return r;   // split-K partial tiles of the decode GEMMs
    if ((r = A((void**)&e->samp_part, (size_t)B * 8 * 16 * 4))) return r;       // sampler records [row][part <= 8][16]
    {   // beam search
        const size_t ctx = c.n_text_ctx, lt = e->ld_tok;
        if ((r = A((void**)&e->b_tokens_tmp, (size_t)B * lt * 4))) return r;
        if ((r = A((void**)&e->b_node, (size_t)B * ctx * 4))) return r;
        if ((r = A((void**)&e->b_node_tmp, (size_t)B * ctx * 4))) return r;
        if ((r = A((void**)&e->b_cand_tok, (size_t)B * 9 * 4))) return r;
        if ((r = A((void**)&e->b_cand_score, (size_t)B * 9 * 4))) return r;
        if ((r = A((void**)&e->b_src, (size_t)B * 4))) return r;
        if ((r = A((void**)&e->b_next, (size_t)B * 4))) return r;
        if ((r = A((void**)&e->b_sum_tmp, (size_t)B * 4))) return r;
        if ((r = A((void**)&e->b_fin_tokens, (size_t)2 * B * lt * 4))) return r;
        if ((r = A((void**)&e->b_fin_score, (size_t)2 * B * 4))) return r;
        if ((r = A((void**)&e->b_fin_count, (size_t)B * 4))) return r;
    }
    if ((r = A((void**)&e->rs_taps, (size_t)RS_MAX_TAPS * 4))) return r;
    HIP_TRY(hipHostMalloc((void**)&e->h_pinned, 64));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" {

int wx_finalize(wx_engine* e) {
    if (!e) return fail(WX_ERR_INVALID, "null engine");
    if (e->finalized) return fail(WX_ERR_STATE, "engine already finalized");
    if (e->mel_filters_h.empty()) return fail(WX_ERR_STATE, "mel filters not set");
    DeviceScope dev_scope(e);
    int r;
    if ((r = build_weights(e))) return r;
    if ((r = build_workspace(e))) return r;
    e->finalized = true;
    return WX_OK;
}

// A second engine on the SAME weights: the parent's weight pointers are copied, the weight block is shared (freed with the last
// engine that refers to it), workspace / caches / decode state / graph are the new engine's own.
int wx_create_shared(const wx_engine* parent, const wx_config* cfg, wx_engine** out) {
    if (!parent || !cfg || !out) return fail(WX_ERR_INVALID, "null argument");
    if (!parent->finalized) return fail(WX_ERR_STATE, "the parent engine is not finalized");
    const wx_config &c = *cfg, &pc = parent->cfg;
    if (c.n_mels != pc.n_mels || c.n_audio_ctx != pc.n_audio_ctx || c.n_audio_state != pc.n_audio_state || c.n_audio_head != pc.n_audio_head ||
        c.n_audio_layer != pc.n_audio_layer || c.n_vocab != pc.n_vocab || c.n_text_ctx != pc.n_text_ctx || c.n_text_state != pc.n_text_state ||
        c.n_text_head != pc.n_text_head || c.n_text_layer != pc.n_text_layer || c.precision != pc.precision || c.device != pc.device)
        return fail(WX_ERR_INVALID, "wx_create_shared: model dimensions, precision and device must be the parent's");
    if (c.max_batch <= 0) return fail(WX_ERR_INVALID, "bad dimensions");
    wx_engine* e = nullptr;
    int r = wx_create(cfg, &e);
    if (r) return r;
    if (e->xattn != parent->xattn) {
        wx_destroy(e);
        return fail(WX_ERR_INVALID, "wx_create_shared: this max_batch selects the other cross-attention form (its weight layouts are not in the parent)");
    }
    DeviceScope dev_scope(e);
    e->wb = parent->wb;
    e->owns_weights = false;
    e->conv1_w = parent->conv1_w; e->conv2_w = parent->conv2_w; e->conv1_b = parent->conv1_b; e->conv2_b = parent->conv2_b; e->enc_pos = parent->enc_pos;
    e->enc = parent->enc; e->dec = parent->dec;
    e->lnpost_g = parent->lnpost_g; e->lnpost_b = parent->lnpost_b; e->declnf_g = parent->declnf_g; e->declnf_b = parent->declnf_b;
    e->tok_emb = parent->tok_emb; e->dec_pos = parent->dec_pos; e->tok_emb_pk = parent->tok_emb_pk;
    e->mel = parent->mel;
    e->suppress_h = parent->suppress_h;
    const int64_t shared_bytes = e->wb->bytes;
    if ((r = build_workspace(e))) { wx_destroy(e); return r; }
    if (e->wb->bytes != shared_bytes) { wx_destroy(e); return fail(WX_ERR_STATE, "wx_create_shared: the workspace allocated into the shared weight block"); }
    e->weight_bytes = 0;
    e->finalized = true;
    *out = e;
    return WX_OK;
}

}  // extern "C"

extern "C" {

int wx_pcm16_to_f32(wx_engine* e, const int16_t* d_pcm16, int64_t n, float* d_out, void* stream) {
    if (!e || !d_pcm16 || !d_out || n < 0) return fail(WX_ERR_INVALID, "bad argument");
    DeviceScope dev_scope(e);
    WX_TRY(launch_pcm16_to_f32(d_pcm16, d_out, n, (hipStream_t)stream));
    return WX_OK;
}

int wx_resample_pcm16(wx_engine* e, const int16_t* d_pcm16, int64_t n_frames, int32_t n_channels, int32_t up, int32_t down,
                      const float* h_taps, int32_t half_len, int32_t quantize, float* d_out, int64_t n_out, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!d_pcm16 || !h_taps || !d_out || n_frames <= 0) return fail(WX_ERR_INVALID, "bad argument");
    if (n_channels < 1 || n_channels > 8) return fail(WX_ERR_INVALID, "n_channels must be 1..8");
    if (up < 1 || down < 1 || down > 8 * up) return fail(WX_ERR_INVALID, "unsupported rate ratio (down <= 8*up)");
    if (half_len < 0 || 2 * (int64_t)half_len + 1 > RS_MAX_TAPS) return fail(WX_ERR_INVALID, "too many filter taps");
    if (n_out != (n_frames * up + down - 1) / down) return fail(WX_ERR_INVALID, "n_out must be ceil(n_frames*up/down)");
    hipStream_t st = (hipStream_t)stream;
    // the taps are host memory of the caller: the copy is complete when the call returns
    HIP_TRY(hipMemcpyAsync(e->rs_taps, h_taps, (size_t)(2 * half_len + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int r = launch_resample_pcm16(d_pcm16, n_frames, n_channels, up, down, e->rs_taps, half_len, quantize, d_out, n_out, st);
    if (r) return fail(r == -1 ? WX_ERR_INVALID : WX_ERR_HIP, "resample launch failed (filter reach exceeds the LDS span?)");
    return WX_OK;
}

int wx_mel_reset_max(wx_engine* e, float* d_max, void* stream) {
    if (!e || !d_max) return fail(WX_ERR_INVALID, "bad argument");
    DeviceScope dev_scope(e);
    WX_TRY(launch_fill_f32(d_max, -INFINITY, 1, (hipStream_t)stream));
    return WX_OK;
}

int wx_mel_stage1(wx_engine* e, const float* d_pcm, int64_t pcm_offset, int64_t n_valid, int64_t n_file, int64_t n_total,
                  int64_t frame0, int64_t n_frames, float* d_logmel, int64_t ld, float* d_max, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!d_pcm || !d_logmel || !d_max || n_frames < 0 || frame0 < 0 || n_file < 0 || n_total < n_file || n_valid < 0 || ld < n_frames)
        return fail(WX_ERR_INVALID, "bad mel arguments");
    if (n_total < N_FFT) return fail(WX_ERR_INVALID, "audio shorter than one STFT window");
    if ((frame0 + n_frames) * HOP > n_total) return fail(WX_ERR_INVALID, "frame range beyond the padded signal");
    WX_TRY(launch_mel_stage1(e->mel, d_pcm, pcm_offset, n_valid, n_file, n_total, frame0, n_frames, d_logmel, ld, d_max, (hipStream_t)stream));
    return WX_OK;
}

int wx_mel_stage2(wx_engine* e, float* d_logmel, int64_t ld, int64_t n_frames, const float* d_max, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!d_logmel || !d_max || n_frames < 0 || ld < n_frames) return fail(WX_ERR_INVALID, "bad mel arguments");
    WX_TRY(launch_mel_stage2(d_logmel, ld, e->cfg.n_mels, n_frames, d_max, (hipStream_t)stream));
    return WX_OK;
}

int wx_log_mel_static(wx_engine* e, const float* d_pcm, int64_t n_samples, int64_t padding, float* d_logmel, int64_t ld, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!d_pcm || !d_logmel || n_samples <= 0 || padding < 0) return fail(WX_ERR_INVALID, "bad mel arguments");
    const int64_t n_total = n_samples + padding, n_frames = n_total / HOP;
    if (n_total < N_FFT) return fail(WX_ERR_INVALID, "audio shorter than one STFT window");
    if (ld < n_frames) return fail(WX_ERR_INVALID, "bad mel arguments (ld < frame count)");
    const int64_t max_frames = (int64_t)(WX_STATIC_FLOATS - WX_STATIC_HEAD) * 8;      // 8 frames per stage-1 workgroup
    if (n_frames > max_frames)
        return fail(WX_ERR_INVALID, "wx_log_mel_static: " + std::to_string(n_frames) + " frames exceed the static scratch (" +
                                    std::to_string(max_frames) + " frames = 87 min of audio including padding); use wx_mel_stage1/2");
    float* ptr = e->static_scratch;                      // this device's copy of `scratch`
    if (!ptr) return fail(WX_ERR_STATE, "static scratch address was not resolved");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = mel_stage1_blocks(n_frames);
    if (blocks > WX_STATIC_FLOATS - WX_STATIC_HEAD) return fail(WX_ERR_INVALID, "wx_log_mel_static: grid exceeds the static scratch");
    // nothing has to start at zero: the launch below overwrites slots [HEAD, HEAD + blocks), the fold reads exactly those
    WX_TRY(launch_mel_stage1(e->mel, d_pcm, 0, n_samples, n_samples, n_total, 0, n_frames, d_logmel, ld, ptr, st, ptr + WX_STATIC_HEAD));
    WX_TRY(launch_mel_fold_block_max(ptr + WX_STATIC_HEAD, blocks, ptr, st));
    WX_TRY(launch_mel_stage2(d_logmel, ld, e->cfg.n_mels, n_frames, ptr, st));
    return WX_OK;
}

static int encoder_attention(wx_engine* e, int B, hipStream_t st) {
    const int d = e->d, T = e->T;
    const size_t es = dtype_size(e->dt);
    const int64_t per = (int64_t)B * T * d;    // elements per q/k/v split
    AttnArgs a;
    a.dt = e->dt;
    a.Q = e->qkv; a.q_sb = (int64_t)e->H * T * HEAD_DIM; a.q_sq = HEAD_DIM; a.q_sh = (int64_t)T * HEAD_DIM; a.n_q = T;
    a.K = (char*)e->qkv + per * es; a.V = (char*)e->qkv + 2 * per * es; a.kv_sbh = (int64_t)T * HEAD_DIM; a.n_keys = T;
    a.O = e->att; a.o_sb = (int64_t)T * d; a.o_sq = d; a.o_sh = HEAD_DIM;
    a.B = B; a.H = e->H;
    // WX_NO_FLASH=1 routes bf16 through the streaming kernel too (A/B checks of the MFMA kernel)
    static const bool no_flash = getenv("WX_NO_FLASH") != nullptr;
    if (e->dt == DT_BF16 && !no_flash) return launch_attn_flash_bf16(a, st);
    return launch_attn_rows(a, st);
}

// cross-attention K/V of every decoder layer for the B windows of the last encode (k pre-scaled by head_dim^-0.25)
static int project_cross_kv(wx_engine* e, int B, hipStream_t st) {
    const int d = e->d, T = e->T, dtp = e->dt, M = B * T;
    const size_t es = dtype_size(dtp);
    const size_t cross_layer = (size_t)e->cfg.max_batch * e->Ht * T * HEAD_DIM;
    if (!e->crossK) {      // xattn engines: first use (beam search at width > 1, or more rows than the split-K decode path takes)
        int r;                 // unless wx_reserve_cross_kv made room beforehand
        if ((r = dev_alloc(e, &e->crossK, cross_layer * e->cfg.n_text_layer * es, false))) return r;
        if ((r = dev_alloc(e, &e->crossV, cross_layer * e->cfg.n_text_layer * es, false))) { e->crossK = nullptr; return r; }
    }
    for (int i = 0; i < e->cfg.n_text_layer; ++i) {
        const LayerW& L = e->dec[i];
        GemmArgs g = gemm_basic(e, e->enc_out, d, L.xkv, M, 2 * d, d);
        g.ep.bias = L.xbkv; g.ep.colscale_period = d; g.ep.colscale[0] = e->qk_scale;
        g.ep.mode = 1; g.ep.out_dtype = dtp; g.ep.hs_T = T; g.ep.hs_H = e->Ht; g.ep.hs_d = d;
        g.ep.out_split[0] = (char*)e->crossK + cross_layer * i * es; g.ep.out_split[1] = (char*)e->crossV + cross_layer * i * es;
        WX_TRY(launch_gemm(g, st));
    }
    e->cross_kv_B = B;
    return 0;
}

// The K/V-cache kernels serve: every fp32 engine; on bf16 engines beam search at width > 1 (the beams of a window share one
// K/V stream) and batches beyond the split-K decode path.  Must be called outside graph capture (it may allocate).
static int ensure_cross_kv(wx_engine* e, int rows, int beam_n, hipStream_t st) {
    if (!e->xattn) return 0;                                   // projected by wx_encode
    if (beam_n == 0 || (beam_n == 1 && rows <= DG_MAX_ROWS)) return 0;      // the E-stream path needs no K/V
    if (e->cross_kv_B >= e->encoded_B && e->crossK) return 0;
    return project_cross_kv(e, e->encoded_B, st);
}

int wx_encode(wx_engine* e, const float* d_mel, int64_t win_stride, int64_t row_stride, int64_t n_valid_frames, int32_t B,
              void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!d_mel || B <= 0 || B > e->cfg.max_batch || win_stride <= 0 || row_stride <= 0)
        return fail(WX_ERR_INVALID, "bad encode arguments (B must be 1..max_batch)");
    hipStream_t st = (hipStream_t)stream;
    const int d = e->d, T = e->T, dtp = e->dt;
    const size_t es = dtype_size(dtp);
    const int M = B * T;
    WX_TRY(launch_mel_to_conv_input(d_mel, win_stride, row_stride, n_valid_frames, e->xpad1, dtp, B, e->cfg.n_mels, e->c_pad, st));
    {   // conv1 (k=3, pad 1) + GELU as a GEMM over the padded time-major image
        GemmArgs g = gemm_basic(e, (char*)e->xpad1 + (size_t)0, e->c_pad, e->conv1_w, B * WIN_FRAMES, d, 3 * e->c_pad);
        g.a_rows_per_batch = WIN_FRAMES; g.a_batch_stride = (int64_t)(WIN_FRAMES + 2) * e->c_pad;
        g.ep.bias = e->conv1_b; g.ep.act = 1;
        g.ep.out = e->xpad2; g.ep.out_dtype = dtp; g.ep.ldo = d;
        g.ep.out_rows_per_batch = WIN_FRAMES; g.ep.out_batch_rows = WIN_FRAMES + 2; g.ep.out_row_offset = 1;
        WX_TRY(launch_gemm(g, st));
    }
    {   // conv2 (k=3, stride 2, pad 1) + GELU + positional table
        GemmArgs g = gemm_basic(e, e->xpad2, 2 * d, e->conv2_w, M, d, 3 * d);
        g.a_rows_per_batch = T; g.a_batch_stride = (int64_t)(WIN_FRAMES + 2) * d;
        g.ep.bias = e->conv2_b; g.ep.act = 1; g.ep.rowadd = e->enc_pos; g.ep.rowadd_period = T;
        g.ep.out = e->x; g.ep.out_dtype = DT_F32; g.ep.ldo = d;
        WX_TRY(launch_gemm(g, st));
    }
    const int64_t per = (int64_t)M * d;
    for (int i = 0; i < e->cfg.n_audio_layer; ++i) {
        const LayerW& L = e->enc[i];
        WX_TRY(launch_layernorm(e->x, L.ln1_g, L.ln1_b, e->xn, dtp, M, d, st));
        {
            GemmArgs g = gemm_basic(e, e->xn, d, L.self.wqkv, M, 3 * d, d);
            g.ep.bias = L.self.bqkv; g.ep.colscale_period = d; g.ep.colscale[0] = e->qk_scale; g.ep.colscale[1] = e->qk_scale;
            g.ep.mode = 1; g.ep.out_dtype = dtp; g.ep.hs_T = T; g.ep.hs_H = e->H; g.ep.hs_d = d;
            g.ep.out_split[0] = e->qkv; g.ep.out_split[1] = (char*)e->qkv + per * es; g.ep.out_split[2] = (char*)e->qkv + 2 * per * es;
            WX_TRY(launch_gemm(g, st));
        }
        WX_TRY(encoder_attention(e, B, st));
        {
            GemmArgs g = gemm_basic(e, e->att, d, L.self.wo, M, d, d);
            g.ep.bias = L.self.bo; g.ep.residual = e->x; g.ep.ldr = d; g.ep.out = e->x; g.ep.out_dtype = DT_F32; g.ep.ldo = d;
            WX_TRY(launch_gemm(g, st));
        }
        WX_TRY(launch_layernorm(e->x, L.ln2_g, L.ln2_b, e->xn, dtp, M, d, st));
        {
            GemmArgs g = gemm_basic(e, e->xn, d, L.w1, M, 4 * d, d);
            g.ep.bias = L.b1; g.ep.act = 1; g.ep.out = e->hid; g.ep.out_dtype = dtp; g.ep.ldo = 4 * d;
            WX_TRY(launch_gemm(g, st));
        }
        {
            GemmArgs g = gemm_basic(e, e->hid, 4 * d, L.w2, M, d, 4 * d);
            g.ep.bias = L.b2; g.ep.residual = e->x; g.ep.ldr = d; g.ep.out = e->x; g.ep.out_dtype = DT_F32; g.ep.ldo = d;
            WX_TRY(launch_gemm(g, st));
        }
    }
    WX_TRY(launch_layernorm(e->x, e->lnpost_g, e->lnpost_b, e->enc_out, dtp, M, d, st));
    e->cross_kv_B = 0;
    e->encoded_B = B;
    if (!e->xattn) return project_cross_kv(e, B, st);
    WX_TRY(launch_xattn_pack_e(e->enc_out, e->enc_frag, B, e->T, d, e->x_nt16, st));
    return WX_OK;
}

int wx_get_encoder_output(wx_engine* e, int32_t B, float* d_out, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!d_out || B <= 0 || B > e->encoded_B) return fail(WX_ERR_INVALID, "no encoder output for that batch");
    WX_TRY(launch_to_f32(e->enc_out, e->dt, d_out, (int64_t)B * e->T * e->d, (hipStream_t)stream));
    return WX_OK;
}

int wx_reserve_cross_kv(wx_engine* e) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (e->crossK) return WX_OK;                              // fp32 / small bf16 engines own it since wx_finalize
    const size_t bytes = (size_t)e->cfg.max_batch * e->Ht * e->T * HEAD_DIM * e->cfg.n_text_layer * dtype_size(e->dt);
    int r;
    if ((r = dev_alloc(e, &e->crossK, bytes, false))) { e->crossK = nullptr; return r; }
    if ((r = dev_alloc(e, &e->crossV, bytes, false))) { e->crossK = nullptr; return r; }     // (the first half stays on the engine's free list)
    e->cross_kv_B = 0;
    return WX_OK;
}

int wx_compact_windows(wx_engine* e, const int32_t* h_src, int32_t n, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!h_src || n <= 0 || n > e->encoded_B) return fail(WX_ERR_INVALID, "bad compaction arguments (1 <= n <= windows of the last wx_encode)");
    for (int i = 0; i < n; ++i)
        if (h_src[i] < i || h_src[i] >= e->encoded_B || (i > 0 && h_src[i] <= h_src[i - 1]))
            return fail(WX_ERR_INVALID, "h_src must be strictly ascending slots of the last wx_encode");
    hipStream_t st = (hipStream_t)stream;
    const int d = e->d, T = e->T;
    const size_t es = dtype_size(e->dt);
    const bool kv_valid = e->crossK && (!e->xattn || e->cross_kv_B >= e->encoded_B);
    const size_t enc_slot = (size_t)T * d * es, frag_slot = (size_t)e->x_nt16 * 16 * d * 2, kv_slot = (size_t)e->Ht * T * HEAD_DIM * es;
    const size_t cross_layer = (size_t)e->cfg.max_batch * e->Ht * T * HEAD_DIM * es;
    for (int i = 0; i < n; ++i) {          // ascending and h_src[i] >= i: a slot is read before anything overwrites it
        const int sidx = h_src[i];
        if (sidx == i) continue;
        HIP_TRY(hipMemcpyAsync((char*)e->enc_out + i * enc_slot, (char*)e->enc_out + sidx * enc_slot, enc_slot, hipMemcpyDeviceToDevice, st));
        if (e->xattn) HIP_TRY(hipMemcpyAsync((char*)e->enc_frag + i * frag_slot, (char*)e->enc_frag + sidx * frag_slot, frag_slot, hipMemcpyDeviceToDevice, st));
        if (kv_valid)
            for (int l = 0; l < e->cfg.n_text_layer; ++l) {
                HIP_TRY(hipMemcpyAsync((char*)e->crossK + l * cross_layer + i * kv_slot, (char*)e->crossK + l * cross_layer + sidx * kv_slot, kv_slot, hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemcpyAsync((char*)e->crossV + l * cross_layer + i * kv_slot, (char*)e->crossV + l * cross_layer + sidx * kv_slot, kv_slot, hipMemcpyDeviceToDevice, st));
            }
    }
    e->cross_kv_B = kv_valid ? n : 0;
    e->encoded_B = n;
    return WX_OK;
}

// self-attention cache addressing for rows b0..: greedy = pages of 16 positions owned by the sequence; beam search = one
// node per token, listed per row in e->b_node (following the beams gathers index rows, never K/V)
static void set_self_cache(const wx_engine* e, AttnArgs& a, int b0) {
    if (e->beam_n > 0) { a.page_table = e->b_node + (size_t)b0 * e->cfg.n_text_ctx; a.pt_ld = e->cfg.n_text_ctx; a.page_size = 1; }
    else { a.page_table = e->page_table + (size_t)b0 * e->n_pages_seq; a.pt_ld = e->n_pages_seq; a.page_size = KV_PAGE; }
}

// one decoder step for position *d_pos (all launches read the position from device memory -> graph-replayable)
// fp32 ("token-exact") sequence: every GEMM applies its own epilogue, LayerNorm is a separate kernel.
// Key split of the decoder cross-attention (workgroups per (window, head)).  A function of the ENGINE (max_batch, beam
// width), never of the batch at hand: the split fixes the order in which partial softmax states are merged, so with a
// per-call choice a window's bf16 logits -- and at a near-tie its token ids -- would depend on how many other windows
// happened to share its batch.
static int cross_attn_key_split(const wx_engine* e) {
    const int windows = e->cfg.max_batch / (e->beam_n > 0 ? e->beam_n : 1);
    int n_split = 1;
    while (n_split < 6 && (windows > 0 ? windows : 1) * e->Ht * n_split < 512) ++n_split;
    return n_split;
}

static int decode_step_f32(wx_engine* e, int B, const SamplerArgs* sa, hipStream_t st) {
    const int d = e->d, dtp = e->dt, T = e->T;
    const size_t es = dtype_size(dtp);
    const wx_config& c = e->cfg;
    WX_TRY(launch_embed(e->tok_emb, e->dec_pos, dtp, e->tokens, e->ld_tok, e->d_pos, e->dx, B, d, st, 1, e->cfg.n_vocab, e->cfg.n_text_ctx));
    const size_t cross_layer = (size_t)c.max_batch * e->Ht * T * HEAD_DIM;
    const size_t self_layer = (size_t)c.max_batch * e->n_pages_seq * e->Ht * KV_PAGE * HEAD_DIM;
    const int n_split = cross_attn_key_split(e);
    for (int i = 0; i < c.n_text_layer; ++i) {
        const LayerW& L = e->dec[i];
        WX_TRY(launch_layernorm(e->dx, L.ln1_g, L.ln1_b, e->dxn, dtp, B, d, st));
        {
            GemmArgs g = gemm_basic(e, e->dxn, d, L.self.wqkv, B, 3 * d, d);
            g.ep.bias = L.self.bqkv; g.ep.colscale_period = d; g.ep.colscale[0] = e->qk_scale; g.ep.colscale[1] = e->qk_scale;
            g.ep.out = e->dqkv; g.ep.out_dtype = dtp; g.ep.ldo = 3 * d;
            WX_TRY(launch_gemm(g, st));
        }
        {
            AttnArgs a;
            a.dt = dtp; a.Q = e->dqkv; a.q_sb = 3 * d; a.n_q = 1;
            set_self_cache(e, a, 0);
            a.Kpool = (char*)e->selfK + self_layer * i * es; a.Vpool = (char*)e->selfV + self_layer * i * es;
            a.Knew = (char*)e->dqkv + (size_t)d * es; a.Vnew = (char*)e->dqkv + (size_t)2 * d * es; a.new_sb = 3 * d;
            a.d_pos = e->d_pos; a.O = e->datt; a.o_sb = d; a.B = B; a.H = e->Ht; a.n_split = 1;
            WX_TRY(launch_attn_decode(a, st));
        }
        {
            GemmArgs g = gemm_basic(e, e->datt, d, L.self.wo, B, d, d);
            g.ep.bias = L.self.bo; g.ep.residual = e->dx; g.ep.ldr = d; g.ep.out = e->dx; g.ep.out_dtype = DT_F32; g.ep.ldo = d;
            WX_TRY(launch_gemm(g, st));
        }
        WX_TRY(launch_layernorm(e->dx, L.lnx_g, L.lnx_b, e->dxn, dtp, B, d, st));
        {
            GemmArgs g = gemm_basic(e, e->dxn, d, L.xq, B, d, d);
            g.ep.bias = L.xbq; g.ep.colscale_period = d; g.ep.colscale[0] = e->qk_scale;
            g.ep.out = e->dq; g.ep.out_dtype = dtp; g.ep.ldo = d;
            WX_TRY(launch_gemm(g, st));
        }
        {
            AttnArgs a;
            a.dt = dtp; a.Q = e->dq; a.q_sb = d; a.n_q = 1;
            a.K = (char*)e->crossK + cross_layer * i * es; a.V = (char*)e->crossV + cross_layer * i * es;
            a.kv_sbh = (int64_t)T * HEAD_DIM; a.n_keys = T; a.kv_group = e->beam_n > 0 ? e->beam_n : 1;
            a.O = e->datt; a.o_sb = d; a.B = B; a.H = e->Ht; a.n_split = n_split; a.partial = e->partial;
            const bool prof = e->prof_on && e->prof_used + 2 <= e->prof_events.size();
            if (prof) HIP_TRY(hipEventRecord(e->prof_events[e->prof_used], st));
            WX_TRY(launch_attn_decode(a, st));
            if (prof) { HIP_TRY(hipEventRecord(e->prof_events[e->prof_used + 1], st)); e->prof_used += 2; }
        }
        {
            GemmArgs g = gemm_basic(e, e->datt, d, L.xo, B, d, d);
            g.ep.bias = L.xbo; g.ep.residual = e->dx; g.ep.ldr = d; g.ep.out = e->dx; g.ep.out_dtype = DT_F32; g.ep.ldo = d;
            WX_TRY(launch_gemm(g, st));
        }
        WX_TRY(launch_layernorm(e->dx, L.ln2_g, L.ln2_b, e->dxn, dtp, B, d, st));
        {
            GemmArgs g = gemm_basic(e, e->dxn, d, L.w1, B, 4 * d, d);
            g.ep.bias = L.b1; g.ep.act = 1; g.ep.out = e->dhid; g.ep.out_dtype = dtp; g.ep.ldo = 4 * d;
            WX_TRY(launch_gemm(g, st));
        }
        {
            GemmArgs g = gemm_basic(e, e->dhid, 4 * d, L.w2, B, d, 4 * d);
            g.ep.bias = L.b2; g.ep.residual = e->dx; g.ep.ldr = d; g.ep.out = e->dx; g.ep.out_dtype = DT_F32; g.ep.ldo = d;
            WX_TRY(launch_gemm(g, st));
        }
    }
    WX_TRY(launch_layernorm(e->dx, e->declnf_g, e->declnf_b, e->dxn, dtp, B, d, st));
    {
        GemmArgs g = gemm_basic(e, e->dxn, d, e->tok_emb, B, e->vocab_pad, d);
        g.ep.out = e->logits; g.ep.out_dtype = DT_F32; g.ep.ldo = e->vocab_pad;
        WX_TRY(launch_gemm(g, st));
    }
    if (sa) WX_TRY(launch_sampler(*sa, st));
    return 0;
}

// bf16 sequence (rows [b0, b0 + B) of the batch, B <= DG_MAX_ROWS): the q|k|v projection, the three residual GEMMs of a layer and
// the cross-attention query projection split K over workgroups and hand raw fp32 partial tiles to their consumer -- the
// next LayerNorm (reduce + bias + residual + LN in one kernel) or the attention kernel's q/k/v load -- so every weight
// byte streams once with the whole chip pulling, and no launch is added.  10 launches per layer (11 where the MLP up projection has to take the split-K path).
// G > 1 = prompt pass: the B rows are G consecutive forced positions (*d_pos ...) of B / G sequences, row r = sequence r / G,
// position r % G.  Row-wise kernels do not care; self-attention appends all new k/v rows first and then reads the cache
// causally per row, cross-attention shares a window's K/V stream among its G rows (the beam-search grouping).
constexpr int ACOMB_FUSE_ROWS = 2;
static int decode_layers_bf16(wx_engine* e, int b0, int B, float* gpartial, float* apartial, bool prof_ok, hipStream_t st,
                              int G = 1, bool want_logits = true) {
    const int d = e->d, dtp = e->dt, T = e->T;
    const size_t es = dtype_size(dtp);
    const wx_config& c = e->cfg;
    float* dx = e->dx + (size_t)b0 * d;
    char* dxn = (char*)e->dxn + (size_t)b0 * d * es;
    char* datt = (char*)e->datt + (size_t)b0 * d * es;
    char* dhid = (char*)e->dhid + (size_t)b0 * 4 * d * es;
    if (G < 1 || B % G || (G > 1 && (e->beam_n > 0 || b0 != 0))) return fail(WX_ERR_INVALID, "bad prompt-pass grouping");
    WX_TRY(launch_embed(e->tok_emb, e->dec_pos, dtp, e->tokens + (size_t)b0 * e->ld_tok, e->ld_tok, e->d_pos, dx, B, d, st, G, c.n_vocab, c.n_text_ctx));
    const size_t cross_layer = (size_t)c.max_batch * e->Ht * T * HEAD_DIM, cross_b0 = (size_t)b0 * e->Ht * T * HEAD_DIM;
    const size_t self_layer = (size_t)c.max_batch * e->n_pages_seq * e->Ht * KV_PAGE * HEAD_DIM;
    const int n_split = cross_attn_key_split(e);
    // tuning hooks (WX_SPLIT_DD / FC2 / QKV / FC1 override the K-split of the four decode GEMM shapes)
    static const int o_dd = getenv("WX_SPLIT_DD") ? atoi(getenv("WX_SPLIT_DD")) : 0, o_fc2 = getenv("WX_SPLIT_FC2") ? atoi(getenv("WX_SPLIT_FC2")) : 0,
                     o_qkv = getenv("WX_SPLIT_QKV") ? atoi(getenv("WX_SPLIT_QKV")) : 0, o_fc1 = getenv("WX_SPLIT_FC1") ? atoi(getenv("WX_SPLIT_FC1")) : 0;
    // d x d projections: the split that fills the chip (10 at d = 1280) on the K/V path; 4 on the encoder-output path, where the
    // q~ launch and the reduce + LayerNorm launches read every partial tile again (decode 874-886 -> 863 ms; 507 -> 509 at batch 16 on K/V)
    const int s_dd = o_dd ? o_dd : (e->xattn && decode_gemm_split_ok(d, 4) && decode_gemm_split(d, d) > 4) ? 4 : decode_gemm_split(d, d), s_fc2 = o_fc2 ? o_fc2 : decode_gemm_split(d, 4 * d),
              s_qkv = o_qkv ? o_qkv : decode_gemm_split(3 * d, d), s_fc1 = o_fc1 ? o_fc1 : decode_gemm_split(4 * d, d);
    int pend_split = 0; const float* pend_bias = nullptr;       // residual update not yet folded into dx
    auto split_gemm = [&](const void* A, int64_t lda, const void* W, const void* Wpk, int K, int S) {
        GemmArgs g = gemm_basic(e, A, lda, W, B, d, K);
        g.Wpk = Wpk;
        g.n_split = S; g.partial = gpartial;
        if (S == 1) { g.ep.out = gpartial; g.ep.out_dtype = DT_F32; g.ep.ldo = d; }   // one "partial": same consumer path
        return launch_gemm(g, st);
    };
    for (int i = 0; i < c.n_text_layer; ++i) {
        const LayerW& L = e->dec[i];
        WX_TRY(launch_reduce_layernorm(dx, gpartial, pend_split, pend_bias, L.ln1_g, L.ln1_b, dxn, dtp, B, d, st));
        {   // fused q|k|v projection, split K: the self-attention kernel adds the partial tiles, bias and scale itself
            GemmArgs g = gemm_basic(e, dxn, d, L.self.wqkv, B, 3 * d, d);
            g.Wpk = L.self.wqkv_pk;
            g.n_split = s_qkv; g.partial = gpartial;
            if (s_qkv == 1) { g.ep.out = gpartial; g.ep.out_dtype = DT_F32; g.ep.ldo = 3 * d; }
            WX_TRY(launch_gemm(g, st));
        }
        {
            AttnArgs a;
            a.dt = dtp; a.q_partial = gpartial; a.q_bias = L.self.bqkv; a.q_nsplit = s_qkv; a.q_scale = e->qk_scale; a.q_sb = 3 * d; a.n_q = 1;
            set_self_cache(e, a, b0);
            a.Kpool = (char*)e->selfK + self_layer * i * es; a.Vpool = (char*)e->selfV + self_layer * i * es;
            a.d_pos = e->d_pos; a.O = datt; a.o_sb = d; a.B = B; a.H = e->Ht; a.n_split = 1; a.q_group = G;
            if (G > 1) WX_TRY(launch_kv_append(a, st));
            WX_TRY(launch_attn_decode(a, st));
        }
        WX_TRY(split_gemm(datt, d, L.self.wo, L.self.wo_pk, d, s_dd));
        WX_TRY(launch_reduce_layernorm(dx, gpartial, s_dd, L.self.bo, L.lnx_g, L.lnx_b, dxn, dtp, B, d, st));
        WX_TRY(split_gemm(dxn, d, L.xq, L.xq_pk, d, s_dd));
        const bool prof = prof_ok && e->prof_on && e->prof_used + 2 <= e->prof_events.size();
        bool cross_out_done = false;
        if (e->xattn && e->beam_n <= 1) {
            // cross-attention on the encoder output itself (xattn.hip): q~ = s Wk_h^T q_h, one stream over E per row for
            // all heads, then the value projection of the context vectors.  Row r attends window (b0 + r) / G.
            XQTildeArgs qa;
            qa.q_partial = gpartial; qa.q_bias = L.xbq; qa.q_nsplit = s_dd; qa.Wk_pk = (const bf16*)L.xk_pk; qa.qt = (bf16*)e->x_qt;
            qa.scale = e->qk_scale; qa.rows = B; qa.H = e->Ht; qa.d = d;
            WX_TRY(launch_xattn_qtilde(qa, st));
            XAttnArgs xa;
            xa.qt = (const bf16*)e->x_qt; xa.E = (const bf16*)e->enc_frag + (size_t)(b0 / G) * e->x_nt16 * 16 * d; xa.nt16 = e->x_nt16; xa.cpart = (bf16*)e->x_cpart; xa.ml = e->x_ml;
            xa.rows = B; xa.H = e->Ht; xa.d = d; xa.T = T; xa.group = G; xa.n_split = e->x_split;
            if (prof) HIP_TRY(hipEventRecord(e->prof_events[e->prof_used], st));
            WX_TRY(launch_xattn_estream(xa, st));
            if (prof) { HIP_TRY(hipEventRecord(e->prof_events[e->prof_used + 1], st)); e->prof_used += 2; }
            XCProjArgs ca;
            ca.cpart = (const bf16*)e->x_cpart; ca.ml = e->x_ml; ca.n_split = e->x_split;
            ca.Wv_pk = (const bf16*)L.xv_pk; ca.bv = L.xbkv + d; ca.out = (bf16*)datt; ca.ldo = d; ca.rows = B; ca.H = e->Ht; ca.d = d;
            WX_TRY(launch_xattn_cproj(ca, st));
        } else {
            AttnArgs a;
            a.dt = dtp; a.q_partial = gpartial; a.q_bias = L.xbq; a.q_nsplit = s_dd; a.q_scale = e->qk_scale; a.q_sb = d; a.n_q = 1;
            a.K = (char*)e->crossK + (cross_layer * i + cross_b0) * es; a.V = (char*)e->crossV + (cross_layer * i + cross_b0) * es;
            a.kv_sbh = (int64_t)T * HEAD_DIM; a.n_keys = T; a.kv_group = e->beam_n > 0 ? e->beam_n : G;
            a.O = datt; a.o_sb = d; a.B = B; a.H = e->Ht; a.n_split = n_split; a.partial = apartial;
            // one or two rows: the key splits are combined in the prologue of the output projection (gemm.hip, launch_gemm_acomb)
            // instead of a launch of their own (1.50 instead of 1.56 ms per position at one row; at four rows the prologue's loads
            // cost more than the launch: 1.69 against 1.67).  WX_NO_FUSE_COMBINE=1: the stand-alone combine (read per eager pass / capture: tests toggle it)
            a.defer_combine = n_split > 1 && B <= ACOMB_FUSE_ROWS && a.kv_group == 1 && dtp == DT_BF16 && L.xo_pk && getenv("WX_NO_FUSE_COMBINE") == nullptr;
            if (prof) HIP_TRY(hipEventRecord(e->prof_events[e->prof_used], st));
            WX_TRY(launch_attn_decode(a, st));
            if (prof) { HIP_TRY(hipEventRecord(e->prof_events[e->prof_used + 1], st)); e->prof_used += 2; }
            if (a.defer_combine) {
                GemmArgs g = gemm_basic(e, datt, d, L.xo, B, d, d);
                g.Wpk = L.xo_pk;
                g.n_split = s_dd; g.partial = gpartial;
                if (s_dd == 1) { g.ep.out = gpartial; g.ep.out_dtype = DT_F32; g.ep.ldo = d; }
                g.ac_partial = apartial; g.ac_nsplit = n_split; g.ac_H = e->Ht;
                const int rc = launch_gemm_acomb(g, st);
                if (rc < 0) return fail(WX_ERR_HIP, "launch_gemm_acomb");
                if (rc == 0) cross_out_done = true;
                else WX_TRY(launch_attn_combine(a, st));
            }
        }
        if (!cross_out_done) WX_TRY(split_gemm(datt, d, L.xo, L.xo_pk, d, s_dd));
        WX_TRY(launch_reduce_layernorm(dx, gpartial, s_dd, L.xbo, L.ln2_g, L.ln2_b, dxn, dtp, B, d, st));
        {
            GemmArgs g = gemm_basic(e, dxn, d, L.w1, B, 4 * d, d);
            g.Wpk = L.w1_pk;
            // K halved inside the workgroup, bias + GELU in the epilogue: no partial tiles, no reduce launch
            static const bool no_khalf = getenv("WX_NO_KHALF") != nullptr;
            int fused = 1;
            if (!no_khalf && !o_fc1 && g.Wpk) {
                GemmArgs f = g;
                f.ep.bias = L.b1; f.ep.act = 1; f.ep.out = dhid; f.ep.out_dtype = dtp; f.ep.ldo = 4 * d;
                fused = launch_gemm_khalf(f, st);
                if (fused < 0) return fail(WX_ERR_INVALID, "launch_gemm_khalf refused the MLP up projection");
            }
            if (fused != 0) {
                if (s_fc1 > 1) { g.n_split = s_fc1; g.partial = gpartial; }
                else { g.ep.bias = L.b1; g.ep.act = 1; g.ep.out = dhid; g.ep.out_dtype = dtp; g.ep.ldo = 4 * d; }
                WX_TRY(launch_gemm(g, st));
                if (s_fc1 > 1) WX_TRY(launch_reduce_bias_gelu(gpartial, s_fc1, L.b1, dhid, B, 4 * d, st));
            }
        }
        WX_TRY(split_gemm(dhid, 4 * d, L.w2, L.w2_pk, 4 * d, s_fc2));
        pend_split = s_fc2; pend_bias = L.b2;
    }
    if (!want_logits) return 0;      // prompt pass with no row at its SOT position: the residual stream is not needed again
    WX_TRY(launch_reduce_layernorm(dx, gpartial, pend_split, pend_bias, e->declnf_g, e->declnf_b, dxn, dtp, B, d, st));
    {
        GemmArgs g = gemm_basic(e, dxn, d, e->tok_emb, B, e->vocab_pad, d);
        g.Wpk = e->tok_emb_pk;
        g.ep.out = e->logits + (size_t)b0 * e->vocab_pad; g.ep.out_dtype = DT_F32; g.ep.ldo = e->vocab_pad;
        WX_TRY(launch_gemm(g, st));
    }
    return 0;
}

// More rows than the split-K kernels take at once (DG_MAX_ROWS) run as consecutive passes of e->pass_rows rows each -- the rows of
// a batch do not interact before the sampler, and the pass size is a function of max_batch only -- so a large max_batch costs its
// share of launches, not a fall to the general kernels.
static int decode_step_bf16(wx_engine* e, int B, const SamplerArgs* sa, hipStream_t st) {
    for (int b0 = 0; b0 < B; b0 += e->pass_rows) {
        const int rows = B - b0 < e->pass_rows ? B - b0 : e->pass_rows;
        int r = decode_layers_bf16(e, b0, rows, e->gpartial, e->partial, b0 == 0, st);
        if (r) return r;
    }
    if (sa) WX_TRY(launch_sampler(*sa, st));
    return 0;
}

// decoder layers for B rows at position *d_pos, then the greedy sampler (sa) or nothing (the beam step follows).
// bf16 split-K path: up to DG_MAX_ROWS rows per pass; greedy / sampling batches beyond that in several passes; beam search
// beyond that (its rows share K/V streams and cache nodes across the whole batch) on the general kernels.
static int decode_step(wx_engine* e, int B, const SamplerArgs* sa, hipStream_t st) {
    return (e->dt == DT_BF16 && (B <= DG_MAX_ROWS || e->beam_n == 0)) ? decode_step_bf16(e, B, sa, st) : decode_step_f32(e, B, sa, st);
}

static SamplerArgs make_sampler(wx_engine* e, int B, const wx_decode_opts& o, int forced) {
    const wx_config& c = e->cfg;
    SamplerArgs s;
    s.logits = e->logits; s.ldl = e->vocab_pad; s.n_vocab = c.n_vocab;
    s.tokens = e->tokens; s.ld_tok = e->ld_tok; s.d_pos = e->d_pos; s.d_block_counter = e->d_counter;
    s.suppress = e->suppress; s.sum_logprob = e->sum_logprob; s.no_speech = e->no_speech; s.n_finished = e->n_finished;
    s.B = B; s.n_prompt = o.n_prompt; s.sot_index = o.sot_index;
    s.tok_eot = c.tok_eot; s.tok_blank = c.tok_blank; s.tok_no_speech = c.tok_no_speech;
    s.tok_no_timestamps = c.tok_no_timestamps; s.tok_timestamp_begin = c.tok_timestamp_begin;
    s.without_timestamps = o.without_timestamps; s.suppress_blank = o.suppress_blank; s.max_initial_ts = o.max_initial_timestamp_index;
    s.forced = forced;
    if (!forced) {
        s.row_n_prompt = e->row_n_prompt; s.row_sot = e->row_sot; s.dyn = e->dyn;
        static const int parts_env = getenv("WX_SAMPLER_PARTS") ? atoi(getenv("WX_SAMPLER_PARTS")) : 4;
        if ((parts_env == 2 || parts_env == 4 || parts_env == 8) && B <= c.max_batch) { s.parts = parts_env; s.part_buf = e->samp_part; }
    }
    return s;
}

// Sampling decode of B rows with per-row prompts: h_rows [B][n_text_ctx + 1] (prompt ids, rest ignored), h_np / h_sot [B].
// temperature 0 = greedy.  Shared by wx_decode_greedy (one prompt for every row) and wx_decode_rows.
static int decode_rows_impl(wx_engine* e, int32_t B, const int32_t* h_rows, const int32_t* h_np, const int32_t* h_sot,
                            const wx_decode_opts* opts, float temperature, uint32_t seed, int32_t* d_tokens, int32_t* d_n_tokens,
                            float* d_sum_logprob, float* d_no_speech_prob, int32_t* h_steps, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!h_rows || !h_np || !h_sot || !opts || !d_tokens || !d_n_tokens || !d_sum_logprob || !d_no_speech_prob) return fail(WX_ERR_INVALID, "null argument");
    if (B <= 0 || B > e->encoded_B) return fail(WX_ERR_INVALID, "decode batch exceeds the last wx_encode batch");
    if (!(temperature >= 0.f)) return fail(WX_ERR_INVALID, "temperature must be >= 0");
    const wx_config& c = e->cfg;
    e->beam_n = 0;
    wx_decode_opts o = *opts;
    int max_np = 0;
    for (int b = 0; b < B; ++b) {
        if (h_np[b] <= 0 || h_np[b] >= c.n_text_ctx || h_sot[b] < 0 || h_sot[b] >= h_np[b]) return fail(WX_ERR_INVALID, "bad prompt length / sot index");
        for (int i = 0; i < h_np[b]; ++i) {
            const int32_t id = h_rows[(size_t)b * e->ld_tok + i];
            if (id < 0 || id >= c.n_vocab) return fail(WX_ERR_INVALID, "prompt id out of range");
        }
        if (h_np[b] > max_np) max_np = h_np[b];
    }
    o.n_prompt = max_np; o.sot_index = h_sot[0];
    int sample_len = o.max_new_tokens > 0 ? o.max_new_tokens : c.n_text_ctx / 2;     // per-row cap; rows also stop at n_text_ctx
    if (o.check_interval < 1) o.check_interval = 8;
    StreamScope sc(e, stream);
    hipStream_t st = sc.st;
    if (int rk = ensure_cross_kv(e, B, 0, st)) return rk;
    // initial state
    {
        std::vector<int32_t> rows((size_t)B * e->ld_tok, c.tok_eot);
        for (int b = 0; b < B; ++b) memcpy(&rows[(size_t)b * e->ld_tok], h_rows + (size_t)b * e->ld_tok, (size_t)h_np[b] * 4);
        const uint32_t dyn[3] = {0, seed, (uint32_t)sample_len};
        uint32_t dynw[3] = {dyn[0], dyn[1], dyn[2]};
        memcpy(&dynw[0], &temperature, 4);
        HIP_TRY(hipMemcpyAsync(e->tokens, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->row_n_prompt, h_np, (size_t)B * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->row_sot, h_sot, (size_t)B * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->dyn, dynw, 12, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(e->d_pos, 0, 64, st));
        HIP_TRY(hipMemsetAsync(e->sum_logprob, 0, (size_t)B * 4, st));
        HIP_TRY(hipMemsetAsync(e->no_speech, 0, (size_t)B * 4, st));
        HIP_TRY(hipStreamSynchronize(st));   // rows / dynw are stack-lifetime host buffers
    }
    const SamplerArgs sa = make_sampler(e, B, o, 0);
    int total_steps = max_np - 1 + sample_len;     // the longest prompt is fed, then up to sample_len sampling steps
    if (total_steps > c.n_text_ctx) total_steps = c.n_text_ctx;     // positions 0 .. n_text_ctx - 1; the last one fills id n_text_ctx
    hipGraphExec_t exec = nullptr;
    auto capture = [&](hipStream_t cs, auto&& body, hipGraphExec_t* out) -> int {
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
        int r = body();
        hipError_t ce = hipStreamEndCapture(cs, &graph);
        if (r != 0) { if (graph) (void)hipGraphDestroy(graph); return r; }
        if (ce != hipSuccess) return fail(WX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
        hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) { *out = nullptr; return fail(WX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
        return 0;
    };
    if (o.use_graph) {
        std::vector<int> key = {0, B, o.without_timestamps, o.suppress_blank, o.max_initial_timestamp_index};
        if (!e->graph_exec || e->graph_key != key) {
            if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
            e->graph_key.clear();
            int r = capture(st, [&] { return decode_step(e, B, &sa, st); }, &e->graph_exec);
            if (r) return r;
            e->graph_key = key;
        }
        exec = e->graph_exec;
    }
    int sampled = 0, min_np = max_np;
    for (int b = 0; b < B; ++b) if (h_np[b] < min_np) min_np = h_np[b];
    const bool want_prof = !exec && o.profile_stride > 0;
    if (want_prof && e->prof_events.empty()) {
        e->prof_events.resize((size_t)2 * c.n_text_layer * 32);
        for (auto& ev : e->prof_events) HIP_TRY(hipEventCreate(&ev));
    }
    e->prof_used = 0; e->prof_ms = 0.0; e->prof_launches = 0;
    // Prompt pass (bf16 split-K path): positions 0 .. min_np - 2 are forced for every row and sample nothing, so they are
    // fed G at a time as rows = sequences x G positions (G <= 8, rows <= min(80, max_batch)) instead of one step each:
    // a 224-id previous-text prompt of one file takes 28 passes, not 224 steps.  Row-wise arithmetic is unchanged, so ids
    // and log-probabilities are bit-identical to stepwise feeding (tests/test_gpu_parity.py); WX_NO_PREFILL=1 disables it.
    int s_begin = 0;
    {
        const char* np_env = getenv("WX_NO_PREFILL");
        const int row_cap = c.max_batch < DG_MAX_ROWS ? c.max_batch : DG_MAX_ROWS;
        int G = row_cap / B; if (G > 8) G = 8;
        if (e->dt == DT_BF16 && G >= 2 && !(np_env && np_env[0] == '1')) {
            SamplerArgs pa = make_sampler(e, B, o, 1);
            pa.row_sot = e->row_sot;                   // the no-speech probability is read at each row's own SOT position
            while (min_np - 1 - s_begin >= 2) {
                const int g = (min_np - 1 - s_begin) < G ? (min_np - 1 - s_begin) : G;
                bool sot_here = false;
                for (int b = 0; b < B; ++b) if (h_sot[b] >= s_begin && h_sot[b] < s_begin + g) sot_here = true;
                int r = decode_layers_bf16(e, 0, B * g, e->gpartial, e->partial, false, st, g, sot_here);
                if (r) return r;
                pa.B = B * g; pa.group = g;
                pa.no_speech = sot_here ? e->no_speech : nullptr;
                WX_TRY(launch_sampler(pa, st));        // forced: only the no-speech read and the position += g
                s_begin += g;
            }
        }
    }
    for (int s = s_begin; s < total_steps; ++s) {
        e->prof_on = want_prof && s >= min_np - 1 && ((s - (min_np - 1)) % o.profile_stride == 0);
        if (exec) HIP_TRY(hipGraphLaunch(exec, st));
        else { int r = decode_step(e, B, &sa, st); e->prof_on = false; if (r) return r; }
        if (s >= min_np - 1) {
            ++sampled;
            if (sampled % o.check_interval == 0 && s + 1 < total_steps) {
                HIP_TRY(hipMemcpyAsync(e->h_pinned, e->n_finished, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (e->h_pinned[0] >= B) break;
            }
        }
    }
    WX_TRY(launch_finish_tokens(e->tokens, e->ld_tok, 0, e->row_n_prompt, c.tok_eot, e->d_pos, e->n_tokens, B, st));
    HIP_TRY(hipMemcpyAsync(d_tokens, e->tokens, (size_t)B * e->ld_tok * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_n_tokens, e->n_tokens, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_sum_logprob, e->sum_logprob, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_no_speech_prob, e->no_speech, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    if (h_steps) *h_steps = sampled;
    if (want_prof) {
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t i = 0; i + 1 < e->prof_used; i += 2) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e->prof_events[i], e->prof_events[i + 1]));
            e->prof_ms += ms; ++e->prof_launches;
        }
    }
    return WX_OK;
}

int wx_decode_greedy(wx_engine* e, int32_t B, const int32_t* h_prompt, const wx_decode_opts* opts, int32_t* d_tokens,
                     int32_t* d_n_tokens, float* d_sum_logprob, float* d_no_speech_prob, int32_t* h_steps, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    if (!h_prompt || !opts) return fail(WX_ERR_INVALID, "null argument");
    const int ctx = e->cfg.n_text_ctx, lt = e->ld_tok;
    if (B <= 0 || B > e->cfg.max_batch) return fail(WX_ERR_INVALID, "decode batch exceeds the last wx_encode batch");
    if (opts->n_prompt <= 0 || opts->n_prompt >= ctx || opts->sot_index < 0 || opts->sot_index >= opts->n_prompt)
        return fail(WX_ERR_INVALID, "bad prompt length / sot index");
    std::vector<int32_t> rows((size_t)B * lt, e->cfg.tok_eot), np(B, opts->n_prompt), sot(B, opts->sot_index);
    for (int b = 0; b < B; ++b) memcpy(&rows[(size_t)b * lt], h_prompt, (size_t)opts->n_prompt * 4);
    return decode_rows_impl(e, B, rows.data(), np.data(), sot.data(), opts, 0.f, 0, d_tokens, d_n_tokens, d_sum_logprob, d_no_speech_prob,
                            h_steps, stream);
}

int wx_decode_rows(wx_engine* e, int32_t B, const int32_t* h_prompts, const int32_t* h_n_prompt, const int32_t* h_sot_index,
                   const wx_decode_opts* opts, float temperature, uint32_t seed, int32_t* d_tokens, int32_t* d_n_tokens,
                   float* d_sum_logprob, float* d_no_speech_prob, int32_t* h_steps, void* stream) {
    return decode_rows_impl(e, B, h_prompts, h_n_prompt, h_sot_index, opts, temperature, seed, d_tokens, d_n_tokens, d_sum_logprob,
                            d_no_speech_prob, h_steps, stream);
}

int wx_decode_beam(wx_engine* e, int32_t B, int32_t beam_size, int32_t max_candidates, const int32_t* h_prompt, const wx_decode_opts* opts,
                   int32_t* d_fin_tokens, float* d_fin_sum_logprob, int32_t* d_fin_count, int32_t* d_beam_tokens,
                   float* d_beam_sum_logprob, float* d_no_speech_prob, int32_t* h_steps, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!h_prompt || !opts || !d_fin_tokens || !d_fin_sum_logprob || !d_fin_count || !d_beam_tokens || !d_beam_sum_logprob || !d_no_speech_prob)
        return fail(WX_ERR_INVALID, "null argument");
    const wx_config& c = e->cfg;
    if (beam_size < 1 || beam_size > 8) return fail(WX_ERR_INVALID, "beam_size must be 1..8");
    // the caller computes upstream's round(beam_size * patience) (Python: double arithmetic, round-half-even) and sizes its
    // finished-list buffers with it; recomputing it here in float would disagree at ties (beam 5, patience 1.7 -> 8 vs 9)
    if (max_candidates < beam_size || max_candidates > 2 * beam_size)
        return fail(WX_ERR_INVALID, "max_candidates must be in [beam_size, 2 * beam_size] (patience in [1, 2])");
    const int n = beam_size, R = B * n, maxc = max_candidates;
    if (B <= 0 || B > e->encoded_B) return fail(WX_ERR_INVALID, "decode batch exceeds the last wx_encode batch");
    if (R > c.max_batch) return fail(WX_ERR_INVALID, "windows x beam_size exceeds max_batch");
    wx_decode_opts o = *opts;
    if (o.n_prompt <= 0 || o.n_prompt >= c.n_text_ctx || o.sot_index < 0 || o.sot_index >= o.n_prompt)
        return fail(WX_ERR_INVALID, "bad prompt length / sot index");
    for (int i = 0; i < o.n_prompt; ++i) if (h_prompt[i] < 0 || h_prompt[i] >= c.n_vocab) return fail(WX_ERR_INVALID, "prompt id out of range");
    int sample_len = o.max_new_tokens > 0 ? o.max_new_tokens : c.n_text_ctx / 2;
    if (o.n_prompt + sample_len > e->ld_tok) sample_len = e->ld_tok - o.n_prompt;     // upstream stops once a row holds n_text_ctx + 1 ids
    if (o.check_interval < 1) o.check_interval = 8;
    StreamScope sc(e, stream);
    hipStream_t st = sc.st;
    if (int rk = ensure_cross_kv(e, R, n, st)) return rk;
    e->beam_n = n;
    {
        std::vector<int32_t> rows((size_t)R * e->ld_tok, c.tok_eot), node((size_t)R * c.n_text_ctx, 0);
        for (int r = 0; r < R; ++r) { memcpy(&rows[(size_t)r * e->ld_tok], h_prompt, (size_t)o.n_prompt * 4); node[(size_t)r * c.n_text_ctx] = r; }
        HIP_TRY(hipMemcpyAsync(e->tokens, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->b_node, node.data(), node.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(e->d_pos, 0, 64, st));
        HIP_TRY(hipMemsetAsync(e->sum_logprob, 0, (size_t)R * 4, st));
        HIP_TRY(hipMemsetAsync(e->no_speech, 0, (size_t)B * 4, st));
        HIP_TRY(hipMemsetAsync(e->b_fin_count, 0, (size_t)B * 4, st));
        HIP_TRY(hipMemsetAsync(e->b_fin_score, 0, (size_t)B * maxc * 4, st));
        std::vector<int32_t> fin((size_t)B * maxc * e->ld_tok, c.tok_eot);
        HIP_TRY(hipMemcpyAsync(e->b_fin_tokens, fin.data(), fin.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    BeamArgs ba;
    ba.logits = e->logits; ba.ldl = e->vocab_pad; ba.n_vocab = c.n_vocab;
    ba.tokens = e->tokens; ba.tokens_tmp = e->b_tokens_tmp; ba.ld_tok = e->ld_tok; ba.ld_node = c.n_text_ctx;
    ba.node = e->b_node; ba.node_tmp = e->b_node_tmp;
    ba.sum_logprob = e->sum_logprob; ba.sum_tmp = e->b_sum_tmp;
    ba.cand_score = e->b_cand_score; ba.cand_tok = e->b_cand_tok; ba.src = e->b_src; ba.next_tok = e->b_next;
    ba.fin_tokens = e->b_fin_tokens; ba.fin_score = e->b_fin_score; ba.fin_count = e->b_fin_count;
    ba.d_pos = e->d_pos; ba.d_counter = e->d_counter; ba.n_complete = e->n_finished;
    ba.suppress = e->suppress; ba.no_speech = e->no_speech;
    ba.B = B; ba.n = n; ba.max_candidates = maxc; ba.n_prompt = o.n_prompt; ba.sot_index = o.sot_index;
    ba.tok_eot = c.tok_eot; ba.tok_blank = c.tok_blank; ba.tok_no_speech = c.tok_no_speech;
    ba.tok_no_timestamps = c.tok_no_timestamps; ba.tok_timestamp_begin = c.tok_timestamp_begin;
    ba.without_timestamps = o.without_timestamps; ba.suppress_blank = o.suppress_blank; ba.max_initial_ts = o.max_initial_timestamp_index;
    auto step = [&]() -> int {
        int r = decode_step(e, R, nullptr, st);
        if (r) return r;
        WX_TRY(launch_beam_step(ba, st));
        return 0;
    };
    const int total_steps = o.n_prompt - 1 + sample_len;
    hipGraphExec_t exec = nullptr;
    if (o.use_graph) {
        std::vector<int> key = {n, B, o.n_prompt, o.sot_index, o.without_timestamps, o.suppress_blank, o.max_initial_timestamp_index, maxc};
        if (!e->graph_exec || e->graph_key != key) {
            if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
            e->graph_key.clear();
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            int r = step();
            hipError_t ce = hipStreamEndCapture(st, &graph);
            if (r != 0) { if (graph) (void)hipGraphDestroy(graph); return r; }
            if (ce != hipSuccess) return fail(WX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
            hipError_t ie = hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) { e->graph_exec = nullptr; return fail(WX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
            e->graph_key = key;
        }
        exec = e->graph_exec;
    }
    int sampled = 0;
    for (int s = 0; s < total_steps; ++s) {
        if (exec) HIP_TRY(hipGraphLaunch(exec, st));
        else { int r = step(); if (r) return r; }
        if (s >= o.n_prompt - 1) {
            ++sampled;
            if (sampled % o.check_interval == 0 && s + 1 < total_steps) {
                HIP_TRY(hipMemcpyAsync(e->h_pinned, e->n_finished, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (e->h_pinned[0] >= B) break;          // every window has max_candidates finished sequences
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(d_fin_tokens, e->b_fin_tokens, (size_t)B * maxc * e->ld_tok * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_fin_sum_logprob, e->b_fin_score, (size_t)B * maxc * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_fin_count, e->b_fin_count, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_beam_tokens, e->tokens, (size_t)R * e->ld_tok * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_beam_sum_logprob, e->sum_logprob, (size_t)R * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_no_speech_prob, e->no_speech, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    if (h_steps) *h_steps = sampled;
    e->beam_n = 0;
    return WX_OK;
}

int wx_get_profile(wx_engine* e, double* total_ms, int64_t* launches) {
    if (!e || !total_ms || !launches) return fail(WX_ERR_INVALID, "null argument");
    *total_ms = e->prof_ms; *launches = e->prof_launches;
    return WX_OK;
}

int wx_decoder_logits(wx_engine* e, int32_t B, const int32_t* d_tokens, int32_t n, float* d_logits, void* stream) {
    if (!e || !e->finalized) return fail(WX_ERR_STATE, "engine not finalized");
    DeviceScope dev_scope(e);
    if (!d_tokens || !d_logits || n <= 0 || n > e->cfg.n_text_ctx) return fail(WX_ERR_INVALID, "bad argument");
    if (B <= 0 || B > e->encoded_B) return fail(WX_ERR_INVALID, "batch exceeds the last wx_encode batch");
    const wx_config& c = e->cfg;
    e->beam_n = 0;
    StreamScope sc(e, stream);
    hipStream_t st = sc.st;
    if (int rk = ensure_cross_kv(e, B, 0, st)) return rk;
    HIP_TRY(hipMemcpy2DAsync(e->tokens, (size_t)e->ld_tok * 4, d_tokens, (size_t)n * 4, (size_t)n * 4, B, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemsetAsync(e->d_pos, 0, 64, st));
    wx_decode_opts o{};
    o.n_prompt = n; o.sot_index = 0; o.without_timestamps = 1; o.suppress_blank = 0; o.max_initial_timestamp_index = -1;
    SamplerArgs sa = make_sampler(e, B, o, 1);
    sa.no_speech = nullptr;
    for (int t = 0; t < n; ++t) {
        int r = decode_step(e, B, &sa, st);
        if (r) return r;
        HIP_TRY(hipMemcpy2DAsync(d_logits + (size_t)t * c.n_vocab, (size_t)n * c.n_vocab * 4, e->logits, (size_t)e->vocab_pad * 4,
                                 (size_t)c.n_vocab * 4, B, hipMemcpyDeviceToDevice, st));
    }
    return WX_OK;
}

int64_t wx_weight_bytes(wx_engine* e) { return e && e->owns_weights ? e->weight_bytes : 0; }
int64_t wx_workspace_bytes(wx_engine* e) { return e ? e->work_bytes : 0; }
int32_t wx_cross_attention_form(wx_engine* e) { return e && e->xattn ? 1 : 0; }
int32_t wx_cross_key_splits(wx_engine* e) { return !e || !e->finalized ? 0 : e->xattn ? e->x_split : cross_attn_key_split(e); }

// one wave that does nothing for `ns` nanoseconds of the GPU's constant-rate clock (100 MHz on gfx950) and leaves: every lane reaches the exit
__global__ void spin_kernel(long long ticks) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    (void)t0;
    const unsigned long long r0 = wall_clock64();
    while ((long long)(wall_clock64() - r0) < ticks) __builtin_amdgcn_s_sleep(8);
}

int wx_stream_spin(void* stream, int32_t microseconds) {
    if (microseconds < 1 || microseconds > 100000) return fail(WX_ERR_INVALID, "1..100000 microseconds");
    int dev = 0, rate_khz = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev));
    if (rate_khz <= 0) rate_khz = 100000;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * rate_khz / 1000);
    HIP_TRY(hipGetLastError());
    return WX_OK;
}

int wx_stream_create_exclusive(int32_t device, void** out_stream) {
    if (!out_stream) return fail(WX_ERR_INVALID, "null argument");
    int prev = 0, n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(WX_ERR_INVALID, "no such device");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0u);
    for (int i = 0; i < prop.multiProcessorCount; ++i) mask[(size_t)i >> 5] |= 1u << (i & 31);
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;
    const hipError_t ce = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    (void)hipSetDevice(prev);
    if (ce != hipSuccess) return fail(WX_ERR_HIP, std::string("hipExtStreamCreateWithCUMask: ") + hipGetErrorString(ce));
    *out_stream = (void*)st;
    return WX_OK;
}

int wx_stream_destroy(void* stream) {
    if (!stream) return WX_OK;
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return WX_OK;
}

}  // extern "C"
