// librced_hip.so -- C ABI (include/rced.h) and host-side runtime of the R-CED / CR-CED forward.
//
// Replaces, for the one hot path, what the reference does with
//   self.pred = self.model(self.input_x)            model_utils/tester.py:69-83
//   self.saver.restore(self.sess, checkpoint)        model_utils/tester.py:36-39
//   self.sess.run(self.pred, {self.input_x: x})      model_utils/tester.py:85-90
// No torch types, no CPU fallback: without a HIP device every compute entry returns RCED_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rced.h"
#include "kernels_generic.h"
#include "rced_internal.h"
#include "rced_spec.h"

using namespace rced;

namespace {

thread_local std::string g_err;

#define fail rced_fail
}  // namespace

int rced_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

namespace {

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(e_ == hipErrorOutOfMemory ? RCED_ERR_ALLOC : RCED_ERR_HIP, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                   \
  } while (0)

struct DeviceGuard {  // run on the model's device, restore the caller's current device after
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(RCED_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(RCED_ERR_ARG, "device %d out of range [0,%d)", device, n);
  return RCED_OK;
}

// Fold inference BatchNorm into the conv (module.py:27-29): w' = w*s, shift = (b-mean)*s+beta,
// s = gamma / sqrt(var + eps).  Done in double, rounded once to fp32.
void fold_layer(const NetSpec& net, int i, const float* p, std::vector<float>* w4, std::vector<float>* shift4,
                int* cout4_out) {
  const LayerSpec& l = net.layer[i];
  const int cin = layer_cin(net, i), cout = l.cout, cout4 = (cout + 3) & ~3;
  const size_t kelems = (size_t)l.kh * l.kw * cin;
  const float* kernel = p;
  const float* bias = p + kelems * cout;
  const float* bn = l.use_norm ? bias + cout : nullptr;
  w4->assign(kelems * cout4, 0.f);
  shift4->assign(cout4, 0.f);
  for (int c = 0; c < cout; ++c) {
    double s = 1.0, sh = bias[c];
    if (bn) {
      const double g = bn[c], be = bn[cout + c], mu = bn[2 * cout + c], var = bn[3 * cout + c];
      s = g / std::sqrt(var + (double)kBnEps);
      sh = ((double)bias[c] - mu) * s + be;
    }
    (*shift4)[c] = (float)sh;
    for (size_t k = 0; k < kelems; ++k) (*w4)[k * cout4 + c] = (float)((double)kernel[k * cout + c] * s);
  }
  *cout4_out = cout4;
}

int launch_generic(const float* x, float* y, const float* w, const float* shift, const float* skip_pre,
                   const float* skip_post, int frames, int T, int F, int cin, int cout, int cout4, int kh,
                   int kw, int use_act, hipStream_t st) {
  const size_t row = (size_t)kh * (F + kw - 1) * cin * sizeof(float);
  if (row > 64 * 1024) return fail(RCED_ERR_ARG, "generic layer needs %zu B of LDS (> 64 KiB)", row);
  if (frames <= 0) return RCED_OK;
  const int fpw = generic_frames_per_wg(F, cout4, kh, row);
  hipLaunchKernelGGL(conv_layer_generic, dim3((frames + fpw - 1) / fpw), dim3(kGenericThreads), row * fpw, st, x, y, w,
                     shift, skip_pre, skip_post, T, F, cin, cout, cout4, kh, kw, use_act, (kh - 1) / 2, (kw - 1) / 2,
                     fpw, frames);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// model object
// ---------------------------------------------------------------------------------------------
rced_model::~rced_model() {
  DeviceGuard g(device);
  for (auto& l : layers) {
    if (l.w) (void)hipFree(l.w);
    if (l.shift) (void)hipFree(l.shift);
  }
  if (workspace) (void)hipFree(workspace);
  for (auto st : host_streams) if (st) (void)hipStreamDestroy(st);
  for (auto ev : host_events) if (ev) (void)hipEventDestroy(ev);
  if (stage_x) (void)hipFree(stage_x);
  if (stage_y) (void)hipFree(stage_y);
  if (fused) fused_destroy(this);
  for (auto& e : prof_events) {
    (void)hipEventDestroy(e.start);
    (void)hipEventDestroy(e.stop);
  }
}

namespace {

// Greedy liveness-based slot assignment for the layerwise path: tensor k+1 (output of layer k)
// lives in slot[k]; a slot can be reused once every reader of its tensor has run.
void plan_slots(rced_model* m) {
  const NetSpec& net = *m->net;
  const int L = net.n_layers;
  std::vector<int> last_use(L + 1, -1);
  for (int i = 0; i < L; ++i) {
    const LayerSpec& l = net.layer[i];
    last_use[l.src] = i;
    if (l.skip_pre >= 0) last_use[l.skip_pre] = i;
    if (l.skip_post >= 0) last_use[l.skip_post] = i;
  }
  struct Slot { int cap; int tensor; };
  std::vector<Slot> slots;
  m->slot_of_tensor.assign(L + 1, -1);
  for (int i = 0; i < L - 1; ++i) {  // the last layer writes the caller's y
    int best = -1;
    for (int s = 0; s < (int)slots.size(); ++s)
      if (last_use[slots[s].tensor] < i && (best < 0 || slots[s].cap > slots[best].cap)) best = s;
    if (best < 0) {
      slots.push_back({net.layer[i].cout, i + 1});
      best = (int)slots.size() - 1;
    } else {
      slots[best].cap = std::max(slots[best].cap, net.layer[i].cout);
      slots[best].tensor = i + 1;
    }
    m->slot_of_tensor[i + 1] = best;
  }
  m->slot_ch_offset.assign(slots.size() + 1, 0);
  for (size_t s = 0; s < slots.size(); ++s) m->slot_ch_offset[s + 1] = m->slot_ch_offset[s] + slots[s].cap;
}

int ensure_workspace(rced_model* m, size_t bytes) {
  if (bytes <= m->workspace_bytes) return RCED_OK;
  if (m->workspace) {
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(m->workspace);
    m->workspace = nullptr;
    m->workspace_bytes = 0;
  }
  HIP_TRY(hipMalloc(&m->workspace, bytes));
  m->workspace_bytes = bytes;
  return RCED_OK;
}

// utterances per pass of the layerwise path: keep the live activations around 1 GiB
int layerwise_chunk(const rced_model* m, int N, int T) {
  const size_t per_utt = (size_t)T * kFeatureDim * m->slot_ch_offset.back() * sizeof(float);
  size_t c = per_utt ? ((size_t)1 << 30) / per_utt : (size_t)N;
  if (c < 1) c = 1;
  return (int)std::min<size_t>(c, (size_t)N);
}

int forward_layerwise(rced_model* m, const float* x, float* y, int N, int T, hipStream_t st) {
  const NetSpec& net = *m->net;
  const int F = kFeatureDim, L = net.n_layers;
  const int chunk = layerwise_chunk(m, N, T);
  const size_t px_chunk = (size_t)chunk * T * F;
  if (int rc = ensure_workspace(m, px_chunk * m->slot_ch_offset.back() * sizeof(float))) return rc;
  float* ws = static_cast<float*>(m->workspace);
  for (int n0 = 0; n0 < N; n0 += chunk) {
    const int nb = std::min(chunk, N - n0);
    const float* xin = x + (size_t)n0 * T * F;
    float* yout = y + (size_t)n0 * T * F;
    auto tensor_ptr = [&](int id) -> const float* {
      if (id < 0) return nullptr;
      if (id == 0) return xin;
      return ws + px_chunk * m->slot_ch_offset[m->slot_of_tensor[id]];
    };
    for (int i = 0; i < L; ++i) {
      const LayerSpec& l = net.layer[i];
      const rced_layer_dev& d = m->layers[i];
      float* out = (i == L - 1) ? yout : const_cast<float*>(tensor_ptr(i + 1));
      m->prof_begin(RCED_K_GENERIC, st);
      if (int rc = launch_generic(tensor_ptr(l.src), out, d.w, d.shift, tensor_ptr(l.skip_pre),
                                  tensor_ptr(l.skip_post), nb * T, T, F, d.cin, l.cout, d.cout4, l.kh, l.kw,
                                  l.use_act, st))
        return rc;
      m->prof_end(RCED_K_GENERIC, st);
    }
  }
  return RCED_OK;
}

}  // namespace

void rced_model::prof_begin(int kind, hipStream_t st) {
  if (!profile || prof_events.size() >= 65536) return;
  ProfEvent e;
  e.kind = kind;
  if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess) return;
  (void)hipEventRecord(e.start, st);
  prof_events.push_back(e);
}
void rced_model::prof_end(int kind, hipStream_t st) {
  if (!profile || prof_events.empty()) return;
  ProfEvent& e = prof_events.back();
  if (e.kind == kind) (void)hipEventRecord(e.stop, st);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* rced_version(void) { return "rced-hip 0.1 gfx950"; }
const char* rced_last_error(void) { return g_err.c_str(); }

int rced_num_layers(int variant) {
  const NetSpec* n = net_spec(variant);
  return n ? n->n_layers : -1;
}
size_t rced_num_weights(int variant) {
  const NetSpec* n = net_spec(variant);
  return n ? net_num_weights(*n) : 0;
}
size_t rced_num_trainable(int variant) {
  const NetSpec* n = net_spec(variant);
  return n ? net_num_trainable(*n) : 0;
}
int rced_layer_desc(int variant, int layer, int out[9]) {
  const NetSpec* n = net_spec(variant);
  if (!n || !out || layer < 0 || layer >= n->n_layers) return fail(RCED_ERR_ARG, "bad variant/layer");
  const LayerSpec& l = n->layer[layer];
  const int v[9] = {l.cout, l.kh, l.kw, l.use_norm, l.use_act, l.src, l.skip_pre, l.skip_post, layer_cin(*n, layer)};
  memcpy(out, v, sizeof(v));
  return RCED_OK;
}
const char* rced_layer_scope(int variant, int layer) {
  const NetSpec* n = net_spec(variant);
  if (!n || layer < 0 || layer >= n->n_layers) return nullptr;
  return n->layer[layer].scope;
}

int rced_create(int variant, const float* blob, size_t n_floats, int device, rced_model** out) {
  if (!out) return fail(RCED_ERR_ARG, "out is NULL");
  *out = nullptr;
  const NetSpec* net = net_spec(variant);
  if (!net) return fail(RCED_ERR_ARG, "unknown variant %d (use RCED_V1/V2/V3)", variant);
  if (!blob) return fail(RCED_ERR_ARG, "blob is NULL");
  if (n_floats != net_num_weights(*net))
    return fail(RCED_ERR_ARG, "blob has %zu floats, variant %d needs %zu", n_floats, variant, net_num_weights(*net));
  for (size_t i = 0; i < n_floats; ++i)
    if (!std::isfinite(blob[i])) return fail(RCED_ERR_ARG, "blob[%zu] is not finite", i);
  if (int rc = check_device(device)) return rc;
  DeviceGuard g(device);
  if (!g.ok) return fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RCED_ERR_HIP, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

  rced_model* m = new (std::nothrow) rced_model();
  if (!m) return fail(RCED_ERR_ALLOC, "host allocation failed");
  m->variant = variant;
  m->device = device;
  m->net = net;
  m->num_cus = prop.multiProcessorCount;
  m->host_blob.assign(blob, blob + n_floats);
  m->layers.resize(net->n_layers);
  const float* p = blob;
  for (int i = 0; i < net->n_layers; ++i) {
    std::vector<float> w4, sh4;
    int cout4 = 0;
    // variance must make sqrt(var+eps) real
    const LayerSpec& l = net->layer[i];
    if (l.use_norm) {
      const float* var = p + (size_t)l.kh * l.kw * layer_cin(*net, i) * l.cout + 4 * (size_t)l.cout;
      for (int c = 0; c < l.cout; ++c)
        if (var[c] + kBnEps <= 0.f) {
          delete m;
          return fail(RCED_ERR_ARG, "%s/batch_norm/moving_variance[%d] = %g is not > -eps", l.scope, c, var[c]);
        }
    }
    fold_layer(*net, i, p, &w4, &sh4, &cout4);
    rced_layer_dev& d = m->layers[i];
    d.cin = layer_cin(*net, i);
    d.cout4 = cout4;
    d.host_w = w4;
    d.host_shift = sh4;
    hipError_t e = hipMalloc(&d.w, w4.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d.shift, sh4.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d.w, w4.data(), w4.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d.shift, sh4.data(), sh4.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      delete m;
      return fail(RCED_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e));
    }
    p += layer_num_weights(*net, i);
  }
  plan_slots(m);
  if (int rc = fused_create(m)) {  // packs MFMA weight fragments; leaves m->fused null if unsupported
    delete m;
    return rc;
  }
  *out = m;
  return RCED_OK;
}

void rced_destroy(rced_model* m) { delete m; }

int rced_set_option(rced_model* m, const char* key, int value) {
  if (!m || !key) return fail(RCED_ERR_ARG, "null argument");
  if (!strcmp(key, "path")) {
    if (value < RCED_PATH_AUTO || value > RCED_PATH_FUSED) return fail(RCED_ERR_ARG, "bad path %d", value);
    if (value == RCED_PATH_FUSED && !m->fused) return fail(RCED_ERR_ARG, "no fused path for variant %d", m->variant);
    m->path = value;
    return RCED_OK;
  }
  if (!strcmp(key, "profile")) {  // (re)arms the HIP-event profiler and drops earlier samples
    DeviceGuard g(m->device);
    m->prof_reset();
    m->profile = value != 0;
    return RCED_OK;
  }
  if (!strcmp(key, "host_chunks")) {  // rced_forward_host pipeline depth: 0 = default (8), 1 = no overlap
    if (value < 0 || value > 64) return fail(RCED_ERR_ARG, "host_chunks must be 0..64");
    m->host_chunks = value;
    return RCED_OK;
  }
  {
    DeviceGuard g(m->device);   // some fused options allocate / upload (e.g. "bf16")
    const int rc = fused_set_option(m, key, value);
    if (rc != RCED_OPT_UNKNOWN) return rc;   // RCED_OK, or a failure whose (specific) message the fused runtime has set
  }
  return fail(RCED_ERR_ARG, "unknown option '%s' (or a value it does not take: %d)", key, value);
}

int rced_get_option(rced_model* m, const char* key, int* value) {
  if (!m || !key || !value) return fail(RCED_ERR_ARG, "null argument");
  if (!strcmp(key, "path")) { *value = m->path; return RCED_OK; }
  if (!strcmp(key, "profile")) { *value = m->profile; return RCED_OK; }
  if (!strcmp(key, "has_fused")) { *value = m->fused != nullptr; return RCED_OK; }
  if (!strcmp(key, "num_cus")) { *value = m->num_cus; return RCED_OK; }
  if (!strcmp(key, "host_chunks")) { *value = m->host_chunks; return RCED_OK; }
  if (fused_get_option(m, key, value) == RCED_OK) return RCED_OK;
  return fail(RCED_ERR_ARG, "unknown option '%s'", key);
}

static int use_fused(const rced_model* m) {
  return m->fused && (m->path == RCED_PATH_AUTO || m->path == RCED_PATH_FUSED);
}

int rced_reserve(rced_model* m, int N, int T) {
  if (!m) return fail(RCED_ERR_ARG, "model is NULL");
  if (N < 0 || T < 0) return fail(RCED_ERR_ARG, "negative shape");
  DeviceGuard g(m->device);
  if (!g.ok) return fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", m->device);
  if (N == 0 || T == 0) return RCED_OK;
  if (use_fused(m)) return fused_reserve(m, N, T);
  const int chunk = layerwise_chunk(m, N, T);
  return ensure_workspace(m, (size_t)chunk * T * kFeatureDim * m->slot_ch_offset.back() * sizeof(float));
}

int rced_forward(rced_model* m, const float* x_dev, float* y_dev, int N, int T, void* stream) {
  if (!m) return fail(RCED_ERR_ARG, "model is NULL");
  if (N < 0 || T < 0) return fail(RCED_ERR_ARG, "negative shape N=%d T=%d", N, T);
  if (N == 0 || T == 0) return RCED_OK;  // empty batch: nothing to do (TF returns an empty array)
  if (!x_dev || !y_dev) return fail(RCED_ERR_ARG, "x/y is NULL");
  if ((size_t)N * T > ((size_t)1 << 31) / kFeatureDim * 4) return fail(RCED_ERR_ARG, "N*T too large");
  DeviceGuard g(m->device);
  if (!g.ok) return fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", m->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (use_fused(m)) return fused_forward(m, x_dev, y_dev, N, T, st);
  return forward_layerwise(m, x_dev, y_dev, N, T, st);
}

int rced_forward_host(rced_model* m, const float* x_host, float* y_host, int N, int T) {
  if (!m) return fail(RCED_ERR_ARG, "model is NULL");
  if (N < 0 || T < 0) return fail(RCED_ERR_ARG, "negative shape N=%d T=%d", N, T);
  if (N == 0 || T == 0) return RCED_OK;
  if (!x_host || !y_host) return fail(RCED_ERR_ARG, "x/y is NULL");
  DeviceGuard g(m->device);
  if (!g.ok) return fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", m->device);
  const size_t bytes = (size_t)N * T * kFeatureDim * sizeof(float);
  if (bytes > m->stage_bytes) {
    if (m->stage_x) (void)hipFree(m->stage_x);
    if (m->stage_y) (void)hipFree(m->stage_y);
    m->stage_x = m->stage_y = nullptr;
    m->stage_bytes = 0;
    HIP_TRY(hipMalloc(&m->stage_x, bytes));
    if (hipError_t e = hipMalloc(&m->stage_y, bytes); e != hipSuccess) {
      (void)hipFree(m->stage_x);
      m->stage_x = nullptr;
      return fail(RCED_ERR_ALLOC, "hipMalloc(stage_y, %zu): %s", bytes, hipGetErrorString(e));
    }
    m->stage_bytes = bytes;
  }
  // Small batches: copy in, run, copy out.  Large ones: split the utterances into chunks and overlap the three
  // legs -- this thread uploads chunk i+1 while chunk i computes, a helper thread downloads chunk i-1 (copies
  // from/to pageable numpy memory block their caller, so the two directions need two callers).
  const size_t utt_bytes = (size_t)T * kFeatureDim * sizeof(float);
  int chunks = m->host_chunks > 0 ? m->host_chunks : 8;
  if (chunks > N) chunks = N;
  if (bytes < ((size_t)8 << 20) || chunks <= 1) {
    HIP_TRY(hipMemcpy(m->stage_x, x_host, bytes, hipMemcpyHostToDevice));
    if (int rc = rced_forward(m, (const float*)m->stage_x, (float*)m->stage_y, N, T, nullptr)) return rc;
    HIP_TRY(hipMemcpy(y_host, m->stage_y, bytes, hipMemcpyDeviceToHost));  // synchronises
    return fused_check(m);
  }
  if (!m->host_streams[0]) {
    for (auto& st : m->host_streams) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  }
  hipStream_t s_in = m->host_streams[0], s_run = m->host_streams[1], s_out = m->host_streams[2];
  // workspace for the LARGEST chunk up front: chunks differ by one utterance, and growing it mid-pipeline would
  // synchronise the device and reallocate under the overlap this path exists for
  if (int rc = rced_reserve(m, (N + chunks - 1) / chunks, T)) return rc;
  if ((int)m->host_events.size() < 2 * chunks) {   // events live in the model and are reused by later calls
    const size_t have = m->host_events.size();
    m->host_events.resize(2 * chunks, nullptr);
    for (size_t i = have; i < m->host_events.size(); ++i) {
      const hipError_t e = hipEventCreateWithFlags(&m->host_events[i], hipEventDisableTiming);
      if (e != hipSuccess) {
        m->host_events.resize(i);   // keep what exists (freed with the model)
        return fail(RCED_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
      }
    }
  }
  hipEvent_t* ev_in = m->host_events.data();
  hipEvent_t* ev_done = m->host_events.data() + chunks;
  auto first = [&](int i) { return (int)((long long)N * i / chunks); };
  std::atomic<int> launched{0};
  std::atomic<bool> abort_flag{false};
  hipError_t out_err = hipSuccess;
  std::thread downloader([&] {
    if (hipSetDevice(m->device) != hipSuccess) { out_err = hipErrorInvalidDevice; return; }
    for (int i = 0; i < chunks; ++i) {
      while (launched.load(std::memory_order_acquire) <= i) {
        if (abort_flag.load()) return;
        std::this_thread::yield();
      }
      const size_t off = (size_t)first(i) * utt_bytes, len = (size_t)(first(i + 1) - first(i)) * utt_bytes;
      hipError_t e = hipStreamWaitEvent(s_out, ev_done[i], 0);
      if (e == hipSuccess) e = hipMemcpyAsync((char*)y_host + off, (const char*)m->stage_y + off, len, hipMemcpyDeviceToHost, s_out);
      if (e != hipSuccess) { out_err = e; return; }
    }
    out_err = hipStreamSynchronize(s_out);
  });
  int rc = RCED_OK;
  hipError_t in_err = hipSuccess;
  for (int i = 0; i < chunks && rc == RCED_OK && in_err == hipSuccess; ++i) {
    const int n0 = first(i), n1 = first(i + 1);
    const size_t off = (size_t)n0 * utt_bytes;
    in_err = hipMemcpyAsync((char*)m->stage_x + off, (const char*)x_host + off, (size_t)(n1 - n0) * utt_bytes, hipMemcpyHostToDevice, s_in);
    if (in_err == hipSuccess) in_err = hipEventRecord(ev_in[i], s_in);
    if (in_err == hipSuccess) in_err = hipStreamWaitEvent(s_run, ev_in[i], 0);
    if (in_err != hipSuccess) break;
    rc = rced_forward(m, (const float*)((const char*)m->stage_x + off), (float*)((char*)m->stage_y + off), n1 - n0, T, s_run);
    if (rc == RCED_OK) in_err = hipEventRecord(ev_done[i], s_run);
    if (rc == RCED_OK && in_err == hipSuccess) launched.store(i + 1, std::memory_order_release);
  }
  if (rc != RCED_OK || in_err != hipSuccess) abort_flag.store(true);
  downloader.join();
  (void)hipStreamSynchronize(s_run);
  if (rc != RCED_OK) return rc;
  if (in_err != hipSuccess) return fail(RCED_ERR_HIP, "host path upload: %s", hipGetErrorString(in_err));
  if (out_err != hipSuccess) return fail(RCED_ERR_HIP, "host path download: %s", hipGetErrorString(out_err));
  return fused_check(m);
}

int rced_conv_bn_relu(const float* x, float* y, const float* kernel, const float* bias, const float* bn,
                      const float* skip_input, int use_act, int N, int T, int F, int cin, int cout, int kh,
                      int kw, int device, void* stream) {
  if (N < 0 || T < 0 || F <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0)
    return fail(RCED_ERR_ARG, "bad shape");
  if (N == 0 || T == 0) return RCED_OK;
  if (!x || !y || !kernel || !bias) return fail(RCED_ERR_ARG, "null pointer");
  if (int rc = check_device(device)) return rc;
  DeviceGuard g(device);
  if (!g.ok) return fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // fold on the host: pull the (small) parameters back, fold in double, push the folded copy
  const size_t kelems = (size_t)kh * kw * cin;
  std::vector<float> hk(kelems * cout), hb(cout), hbn(bn ? 4 * cout : 0);
  HIP_TRY(hipMemcpyAsync(hk.data(), kernel, hk.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hb.data(), bias, hb.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  if (bn) HIP_TRY(hipMemcpyAsync(hbn.data(), bn, hbn.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int cout4 = (cout + 3) & ~3;
  std::vector<float> w4(kelems * cout4, 0.f), sh4(cout4, 0.f);
  for (int c = 0; c < cout; ++c) {
    double s = 1.0, sh = hb[c];
    if (bn) {
      if (hbn[3 * cout + c] + kBnEps <= 0.f) return fail(RCED_ERR_ARG, "moving_variance[%d] not > -eps", c);
      s = (double)hbn[c] / std::sqrt((double)hbn[3 * cout + c] + (double)kBnEps);
      sh = ((double)hb[c] - hbn[2 * cout + c]) * s + hbn[cout + c];
    }
    sh4[c] = (float)sh;
    for (size_t k = 0; k < kelems; ++k) w4[k * cout4 + c] = (float)((double)hk[k * cout + c] * s);
  }
  float *dw = nullptr, *dsh = nullptr;
  HIP_TRY(hipMalloc(&dw, w4.size() * sizeof(float)));
  hipError_t e = hipMalloc(&dsh, sh4.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpyAsync(dw, w4.data(), w4.size() * sizeof(float), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dsh, sh4.data(), sh4.size() * sizeof(float), hipMemcpyHostToDevice, st);
  int rc = RCED_OK;
  if (e != hipSuccess) rc = fail(RCED_ERR_HIP, "parameter upload failed: %s", hipGetErrorString(e));
  if (rc == RCED_OK)
    rc = launch_generic(x, y, dw, dsh, skip_input, nullptr, N * T, T, F, cin, cout, cout4, kh, kw, use_act, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(dw);
  if (dsh) (void)hipFree(dsh);
  return rc;
}

int rced_check(rced_model* m) {
  if (!m) return fail(RCED_ERR_ARG, "model is NULL");
  return fused_check(m);   // reads pinned host memory: no device call, no synchronisation
}

float rced_last_kernel_ms(rced_model* m) {
  if (!m) return -1.f;
  return m->prof_dominant_ms(nullptr, nullptr);
}

int rced_profile_query(rced_model* m, int kind, float* total_ms, int* launches) {
  if (!m || !total_ms || !launches) return fail(RCED_ERR_ARG, "null argument");
  DeviceGuard g(m->device);
  if (int rc = fused_check(m)) return rc;
  float tot = 0.f;
  int n = 0;
  for (auto& e : m->prof_events) {
    if (e.kind != kind) continue;
    if (hipEventSynchronize(e.stop) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) { tot += ms; ++n; }
  }
  *total_ms = tot;
  *launches = n;
  return RCED_OK;
}

}  // extern "C"

void rced_model::prof_reset() {
  for (auto& e : prof_events) {
    (void)hipEventDestroy(e.start);
    (void)hipEventDestroy(e.stop);
  }
  prof_events.clear();
}

float rced_model::prof_dominant_ms(int* kind_out, int* launches_out) {
  float tot[RCED_K_COUNT] = {0};
  int cnt[RCED_K_COUNT] = {0};
  for (auto& e : prof_events) {
    if (hipEventSynchronize(e.stop) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) { tot[e.kind] += ms; ++cnt[e.kind]; }
  }
  int best = -1;
  for (int k = 0; k < RCED_K_COUNT; ++k)
    if (cnt[k] && (best < 0 || tot[k] > tot[best])) best = k;
  if (best < 0) return -1.f;
  if (kind_out) *kind_out = best;
  if (launches_out) *launches_out = cnt[best];
  return tot[best] / cnt[best];
}
