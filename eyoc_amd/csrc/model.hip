// ResUNet2 family (ResUNetBN2C in production): weight packing and the forward schedule.
// Mirrors ResUNet2.__init__/forward (model/resunet.py:18-193) and BasicBlockBase.forward
// (model/residual_block.py:37-53) with every batch norm folded into the convolution before it
// (eval mode, model/common.py:4-6) and every ReLU / residual add / concat fused into a conv epilogue.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "spconv.h"

using namespace eyoc;

namespace {

enum MapSel { M_CONV1, M_S1, M_DOWN, M_UP, M_IDENT, M_AFFINE };   // M_AFFINE: a batch norm that stands alone (ResUNetExpanded's norm<i>_2): y = x * scale + shift per channel

struct LayerPlan {
  std::string name;      // conv name in the state_dict ("block2.conv1")
  std::string norm;      // norm folded into it ("" = none)
  int K, cin, cout;
  MapSel map;
  int level;             // table level (fine level for DOWN/UP); rows of the OUTPUT for S1
  int out_level;         // level whose row count is n_out
  int in_buf, in_col, out_buf, out_col, res_buf;  // buffer ids (-1 none)
  int relu, l2norm, has_bias;
  size_t w_off = 0, b_off = 0;  // float offsets into the blob
  size_t w16_off = 0, s_off = 0;  // SPLIT16 copy of the weights (same size) and the layer's out_scale scalar (spconv layers only)
};

// activation buffers
enum Buf { B_IN = 0, B_X1, B_T1, B_CAT1, B_X2, B_T2, B_CAT2, B_X4, B_T4, B_CAT4, B_X8, B_T8, B_Y8, B_D4, B_DT4, B_D2,
           B_DT2, B_D1, B_DT1, B_H,
           B_U1, B_U2, B_U4, B_U8, B_UD4, B_UD2, B_UD1,    // ResUNetExpanded only (width 0 otherwise): the first block's output of a stage
           B_OUT, B_COUNT };

struct BufPlan { int level, width; };

size_t pad64(size_t v) { return (v + 63) / 64 * 64; }

}  // namespace

struct eyoc_model {
  eyoc_model_desc desc;
  std::vector<LayerPlan> layers;
  BufPlan bufs[B_COUNT];
  float* blob = nullptr;
  size_t blob_floats = 0;
  int math = -1;           // -1 automatic, 0 fp32 MFMA, 1 SPLIT16 (eyoc_model_set_math)
  int last_math = 0;       // what the last forward used
  int timing = 0;
  std::vector<hipEvent_t> events;     // two sets of layers + 1 events (eyoc_model_timing_slot)
  int slot = 0;
  int events_valid[2] = {0, 0};
  hipEvent_t progress_event = nullptr;   // eyoc_model_set_progress_event: recorded in front of layer `progress_layer` of every forward
  int progress_layer = -1;
  unsigned int* range = nullptr;      // device: {overflow flag of the forward in flight, max |activation| bits (probe), probe switch, sticky overflow flag} - split16_guard in spconv.h
  int probe = 0;
};

namespace {

void build_plan(const eyoc_model_desc& d, std::vector<LayerPlan>& L, BufPlan* bufs) {
  const int* C = d.channels;
  const int* T = d.tr_channels;
  const int K1 = d.conv1_kernel_size * d.conv1_kernel_size * d.conv1_kernel_size;
  bufs[B_IN] = {0, d.in_channels};
  bufs[B_X1] = {0, C[1]}; bufs[B_T1] = {0, C[1]}; bufs[B_CAT1] = {0, T[2] + C[1]};
  bufs[B_X2] = {1, C[2]}; bufs[B_T2] = {1, C[2]}; bufs[B_CAT2] = {1, T[3] + C[2]};
  bufs[B_X4] = {2, C[3]}; bufs[B_T4] = {2, C[3]}; bufs[B_CAT4] = {2, T[4] + C[3]};
  bufs[B_X8] = {3, C[4]}; bufs[B_T8] = {3, C[4]}; bufs[B_Y8] = {3, C[4]};
  bufs[B_D4] = {2, T[4]}; bufs[B_DT4] = {2, T[4]};
  bufs[B_D2] = {1, T[3]}; bufs[B_DT2] = {1, T[3]};
  bufs[B_D1] = {0, T[2]}; bufs[B_DT1] = {0, T[2]};
  bufs[B_H] = {0, T[1]};
  bufs[B_OUT] = {0, d.out_channels};
  const int ex = d.expanded ? 1 : 0;
  bufs[B_U1] = {0, ex * C[1]}; bufs[B_U2] = {1, ex * C[2]}; bufs[B_U4] = {2, ex * C[3]}; bufs[B_U8] = {3, ex * C[4]};
  bufs[B_UD4] = {2, ex * T[4]}; bufs[B_UD2] = {1, ex * T[3]}; bufs[B_UD1] = {0, ex * T[2]};

  auto conv = [&](const char* name, const char* norm, int K, int cin, int cout, MapSel map, int level, int out_level,
                  int in_buf, int in_col, int out_buf, int out_col, int res_buf, int relu) {
    LayerPlan p;
    p.name = name; p.norm = norm; p.K = K; p.cin = cin; p.cout = cout; p.map = map; p.level = level;
    p.out_level = out_level; p.in_buf = in_buf; p.in_col = in_col; p.out_buf = out_buf; p.out_col = out_col;
    p.res_buf = res_buf; p.relu = relu; p.l2norm = 0; p.has_bias = 0;
    L.push_back(p);
  };
  auto block1 = [&](const std::string& name, int c, int level, int x_buf, int t_buf, int out_buf, int out_col) {
    conv((name + ".conv1").c_str(), (name + ".norm1").c_str(), 27, c, c, M_S1, level, level, x_buf, 0, t_buf, 0, -1, 1);
    conv((name + ".conv2").c_str(), (name + ".norm2").c_str(), 27, c, c, M_S1, level, level, t_buf, 0, out_buf, out_col,
         x_buf, 1);
  };
  // a stage's block(s): ResUNet2 runs block<i>; ResUNetExpanded (model/resunet.py:409-473) runs block<i> -> relu (the block's own
  // last ReLU already) -> norm<i>_2 -> block<i>_2.  The stand-alone norm cannot be folded into a neighbour: its shift reaches
  // block<i>_2.conv1 only through the neighbours a row HAS, and it is the residual of block<i>_2.conv2 - so it is one layer of its own
  // (x -> u by block<i>, u -> x by the norm: x is free once block<i>.conv2 has read it as its residual)
  auto block = [&](const std::string& stage, int c, int level, int x_buf, int t_buf, int u_buf, int out_buf, int out_col) {
    if (!d.expanded) { block1("block" + stage, c, level, x_buf, t_buf, out_buf, out_col); return; }
    block1("block" + stage, c, level, x_buf, t_buf, u_buf, 0);
    conv(("norm" + stage + "_2").c_str(), ("norm" + stage + "_2").c_str(), 0, c, c, M_AFFINE, level, level, u_buf, 0, x_buf, 0, -1, 0);
    block1("block" + stage + "_2", c, level, x_buf, t_buf, out_buf, out_col);
  };
  // encoder (model/resunet.py:143-161)
  conv("conv1", "norm1", K1, d.in_channels, C[1], M_CONV1, 0, 0, B_IN, 0, B_X1, 0, -1, 0);
  block("1", C[1], 0, B_X1, B_T1, B_U1, B_CAT1, T[2]);
  conv("conv2", "norm2", 27, C[1], C[2], M_DOWN, 0, 1, B_CAT1, T[2], B_X2, 0, -1, 0);
  block("2", C[2], 1, B_X2, B_T2, B_U2, B_CAT2, T[3]);
  conv("conv3", "norm3", 27, C[2], C[3], M_DOWN, 1, 2, B_CAT2, T[3], B_X4, 0, -1, 0);
  block("3", C[3], 2, B_X4, B_T4, B_U4, B_CAT4, T[4]);
  conv("conv4", "norm4", 27, C[3], C[4], M_DOWN, 2, 3, B_CAT4, T[4], B_X8, 0, -1, 0);
  block("4", C[4], 3, B_X8, B_T8, B_U8, B_Y8, 0);
  // decoder (model/resunet.py:163-186); ME.cat order is [decoder | skip]
  conv("conv4_tr", "norm4_tr", 27, C[4], T[4], M_UP, 2, 2, B_Y8, 0, B_D4, 0, -1, 0);
  block("4_tr", T[4], 2, B_D4, B_DT4, B_UD4, B_CAT4, 0);
  conv("conv3_tr", "norm3_tr", 27, C[3] + T[4], T[3], M_UP, 1, 1, B_CAT4, 0, B_D2, 0, -1, 0);
  block("3_tr", T[3], 1, B_D2, B_DT2, B_UD2, B_CAT2, 0);
  conv("conv2_tr", "norm2_tr", 27, C[2] + T[3], T[2], M_UP, 0, 0, B_CAT2, 0, B_D1, 0, -1, 0);
  block("2_tr", T[2], 0, B_D1, B_DT1, B_UD1, B_CAT1, 0);
  conv("conv1_tr", "", 1, C[1] + T[2], T[1], M_IDENT, 0, 0, B_CAT1, 0, B_H, 0, -1, 1);
  conv("final", "", 1, T[1], d.out_channels, M_IDENT, 0, 0, B_H, 0, B_OUT, 0, -1, 0);
  L.back().has_bias = 1;
  L.back().l2norm = d.normalize_feature ? 1 : 0;
  size_t off = 0;
  for (auto& p : L) {
    p.w_off = off;
    off += p.map == M_AFFINE ? pad64((size_t)p.cout) : pad64((size_t)p.K * p.cin * p.cout);    // a stand-alone norm: its scale per channel
    p.b_off = off;
    off += pad64((size_t)p.cout);
  }
  // second half of the blob: the SPLIT16 packing of every sparse-conv layer (spconv_wave.hip, MATH = 1)
  for (auto& p : L) {
    if (p.map == M_AFFINE) continue;
    if (p.map != M_CONV1) {
      p.w16_off = off;
      off += pad64((size_t)p.K * p.cin * p.cout);
    }
    p.s_off = off;      // sparse-conv layers: out_scale; the first convolution: {2^sh, 2^-sh} of its MFMA kernels' weight halves
    off += 64;
  }
}

// y = x * scale[c] + shift[c] on fp32 or SPLIT16 rows (the format of its neighbours in the plan); one thread per 4 channels.  SPLIT16
// rows carry the range guard like every other kernel that writes them (spconv.h split16_guard)
template <bool SPLIT>
__global__ void __launch_bounds__(256) k_affine(const float* __restrict__ in, int ld_in, int n, int c, const float* __restrict__ scale,
                                                const float* __restrict__ shift, float* __restrict__ out, int ld_out, unsigned int* range) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int q4 = c / 4;
  const long long r = t / q4;
  const int ch = (int)(t % q4) * 4;
  float mx = 0.0f;
  if (r < n) {
    const float4 s = *reinterpret_cast<const float4*>(scale + ch), b = *reinterpret_cast<const float4*>(shift + ch);
    float4 v = SPLIT ? split16_load4(in + r * ld_in, ch) : *reinterpret_cast<const float4*>(in + r * ld_in + ch);
    v.x = fmaf(v.x, s.x, b.x); v.y = fmaf(v.y, s.y, b.y); v.z = fmaf(v.z, s.z, b.z); v.w = fmaf(v.w, s.w, b.w);
    if (SPLIT) { split16_track(mx, v); split16_store4(out + r * ld_out, ch, v); }
    else *reinterpret_cast<float4*>(out + r * ld_out + ch) = v;
  }
  if (SPLIT) {
    for (int o = 32; o; o >>= 1) mx = split16_merge(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) split16_report(range, mx);
  }
}

int launch_affine(const float* in, int ld_in, int n, int c, const float* scale, const float* shift, float* out, int ld_out, bool split,
                  unsigned int* range, hipStream_t st) {
  EYOC_REQUIRE(c % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && (!split || (c % 32 == 0 && ld_in % 32 == 0 && ld_out % 32 == 0)),
               EYOC_ERR_INVALID, "model: stand-alone norm over %d channels (rows of %d -> %d floats)", c, ld_in, ld_out);
  if (!n) return EYOC_OK;
  const dim3 grid((unsigned)cdiv((long long)n * (c / 4), 256));
  if (split) hipLaunchKernelGGL(k_affine<true>, grid, dim3(256), 0, st, in, ld_in, n, c, scale, shift, out, ld_out, range);
  else hipLaunchKernelGGL(k_affine<false>, grid, dim3(256), 0, st, in, ld_in, n, c, scale, shift, out, ld_out, range);
  EYOC_CHECK_HIP(hipGetLastError());
  return EYOC_OK;
}

size_t plan_blob_floats(const std::vector<LayerPlan>& L) {
  size_t end = 0;
  for (auto& p : L) {
    end = std::max(end, p.b_off + pad64(p.cout));
    if (p.map != M_AFFINE) end = std::max(end, p.s_off + 64);
  }
  return end;
}

const eyoc_layer_params* find_layer(const eyoc_layer_params* layers, int n, const std::string& name) {
  for (int i = 0; i < n; ++i)
    if (layers[i].name && name == layers[i].name) return &layers[i];
  return nullptr;
}

int check_desc(const eyoc_model_desc* d) {
  EYOC_REQUIRE(d, EYOC_ERR_INVALID, "model: NULL desc");
  EYOC_REQUIRE(d->in_channels >= 1 && d->in_channels <= 64, EYOC_ERR_INVALID, "model: in_channels %d", d->in_channels);
  EYOC_REQUIRE(d->expanded == 0 || d->expanded == 1, EYOC_ERR_INVALID, "model: expanded %d not in {0, 1}", d->expanded);
  EYOC_REQUIRE(d->out_channels == 32 || d->out_channels == 64 || d->out_channels == 128, EYOC_ERR_INVALID,
               "model: out_channels %d not in {32,64,128}", d->out_channels);
  EYOC_REQUIRE(d->conv1_kernel_size == 1 || d->conv1_kernel_size == 3 || d->conv1_kernel_size == 5 ||
               d->conv1_kernel_size == 7, EYOC_ERR_INVALID, "model: conv1_kernel_size %d", d->conv1_kernel_size);
  for (int i = 1; i <= 4; ++i) {
    const int c = d->channels[i], t = d->tr_channels[i];
    EYOC_REQUIRE(c == 32 || c == 64 || c == 128 || c == 256, EYOC_ERR_INVALID, "model: CHANNELS[%d] = %d", i, c);
    EYOC_REQUIRE(t == 32 || t == 64 || t == 128 || t == 256, EYOC_ERR_INVALID, "model: TR_CHANNELS[%d] = %d", i, t);
  }
  EYOC_REQUIRE(d->channels[1] <= 128, EYOC_ERR_INVALID, "model: CHANNELS[1] = %d > 128", d->channels[1]);
  return EYOC_OK;
}

}  // namespace

extern "C" {

size_t eyoc_model_blob_floats(const eyoc_model_desc* desc) {
  if (check_desc(desc) != EYOC_OK) return 0;
  std::vector<LayerPlan> L;
  BufPlan bufs[B_COUNT];
  build_plan(*desc, L, bufs);
  return plan_blob_floats(L);
}

// Folds batch norm into every convolution and writes the packed blob (fp32 fragment order + split16 packing + shifts) into
// HOST memory.  Pure host code: no device is touched, so a rank can build (or verify) the blob it broadcasts without a GPU.
int eyoc_model_pack_host(const eyoc_model_desc* desc, const eyoc_layer_params* layers, int n_layers, float* blob_host,
                         size_t blob_floats) {
  EYOC_REQUIRE(layers && blob_host, EYOC_ERR_INVALID, "eyoc_model_pack_host: NULL argument");
  int rc = check_desc(desc);
  if (rc) return rc;
  std::vector<LayerPlan> plan;
  BufPlan bufs[B_COUNT];
  build_plan(*desc, plan, bufs);
  const size_t need = plan_blob_floats(plan);
  EYOC_REQUIRE(blob_floats >= need, EYOC_ERR_INVALID, "eyoc_model_pack_host: blob of %zu floats required, got %zu", need, blob_floats);
  std::fill(blob_host, blob_host + need, 0.0f);
  for (auto& p : plan) {
    if (p.map == M_AFFINE) {                                            // scale / shift of a norm that stands alone (the same fold as below)
      const eyoc_layer_params* bn = find_layer(layers, n_layers, p.norm);
      EYOC_REQUIRE(bn && bn->bn_weight && bn->bn_bias && bn->bn_mean && bn->bn_var && bn->cout == p.cout, EYOC_ERR_INVALID,
                   "eyoc_model_create: norm '%s' missing or wrong width", p.norm.c_str());
      for (int c = 0; c < p.cout; ++c) {
        const float s = bn->bn_weight[c] / std::sqrt(bn->bn_var[c] + desc->bn_eps);
        blob_host[p.w_off + c] = s;
        blob_host[p.b_off + c] = bn->bn_bias[c] - bn->bn_mean[c] * s;
      }
      continue;
    }
    const eyoc_layer_params* cv = find_layer(layers, n_layers, p.name);
    EYOC_REQUIRE(cv && cv->kernel && cv->K == p.K && cv->cin == p.cin && cv->cout == p.cout, EYOC_ERR_INVALID,
                 "eyoc_model_create: layer '%s' missing or shape mismatch (expected K=%d cin=%d cout=%d, got %d %d %d)",
                 p.name.c_str(), p.K, p.cin, p.cout, cv ? cv->K : -1, cv ? cv->cin : -1, cv ? cv->cout : -1);
    std::vector<float> scale(p.cout, 1.0f), shift(p.cout, 0.0f);
    if (!p.norm.empty()) {
      const eyoc_layer_params* bn = find_layer(layers, n_layers, p.norm);
      EYOC_REQUIRE(bn && bn->bn_weight && bn->bn_bias && bn->bn_mean && bn->bn_var && bn->cout == p.cout, EYOC_ERR_INVALID,
                   "eyoc_model_create: norm '%s' missing or wrong width", p.norm.c_str());
      for (int c = 0; c < p.cout; ++c) {
        const float s = bn->bn_weight[c] / std::sqrt(bn->bn_var[c] + desc->bn_eps);
        scale[c] = s;
        shift[c] = bn->bn_bias[c] - bn->bn_mean[c] * s;
      }
    }
    if (p.has_bias && cv->bias)
      for (int c = 0; c < p.cout; ++c) shift[c] += cv->bias[c];
    float* w = blob_host + p.w_off;
    if (p.map == M_CONV1) {  // plain [K][cin][cout] with the scale folded in
      float wmax = 0.0f;
      for (size_t i = 0; i < (size_t)p.K * p.cin * p.cout; ++i) {
        w[i] = cv->kernel[i] * scale[i % p.cout];
        if (std::isfinite(w[i])) wmax = std::max(wmax, std::fabs(w[i]));
      }
      int e = 1;                                        // wmax = f 2^e, f in [0.5, 1): wmax 2^(9 - e) lies in [256, 512)
      if (wmax > 0.0f) (void)std::frexp(wmax, &e);
      const int sh = std::min(24, std::max(-6, 9 - e));   // same clamp as eyoc_spconv_pack_weights_split16
      blob_host[p.s_off] = std::ldexp(1.0f, sh);
      blob_host[p.s_off + 1] = std::ldexp(1.0f, -sh);
    } else {
      rc = eyoc_spconv_pack_weights(cv->kernel, scale.data(), p.K, p.cin, p.cout, w);
      if (!rc) rc = eyoc_spconv_pack_weights_split16(cv->kernel, scale.data(), p.K, p.cin, p.cout, blob_host + p.w16_off,
                                                     blob_host + p.s_off);
      if (rc) return rc;
    }
    memcpy(blob_host + p.b_off, shift.data(), p.cout * sizeof(float));
  }
  return EYOC_OK;
}

int eyoc_model_create(eyoc_ctx* ctx, const eyoc_model_desc* desc, const eyoc_layer_params* layers, int n_layers,
                      float* blob_dev, size_t blob_floats, eyoc_model** out) {
  EYOC_REQUIRE(ctx && out && blob_dev, EYOC_ERR_INVALID, "eyoc_model_create: NULL argument");
  int rc = check_desc(desc);
  if (rc) return rc;
  eyoc_model* m = new eyoc_model();
  m->desc = *desc;
  build_plan(*desc, m->layers, m->bufs);
  m->blob = blob_dev;
  m->blob_floats = plan_blob_floats(m->layers);
  if (blob_floats < m->blob_floats || ((uintptr_t)blob_dev & 255) != 0) {
    set_error("eyoc_model_create: blob of %zu floats (256-byte aligned) required, got %zu at %p", m->blob_floats,
              blob_floats, (void*)blob_dev);
    delete m;
    return EYOC_ERR_INVALID;
  }
  if (layers) {
    std::vector<float> host(m->blob_floats, 0.0f);
    rc = eyoc_model_pack_host(desc, layers, n_layers, host.data(), host.size());
    if (rc) { delete m; return rc; }
    hipError_t e = hipMemcpy(blob_dev, host.data(), m->blob_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      set_error("eyoc_model_create: weight upload failed: %s", hipGetErrorString(e));
      delete m;
      return EYOC_ERR_HIP;
    }
  }
  {   // the SPLIT16 range-guard words (16 bytes; part of the handle like the timing events)
    hipError_t e = hipMalloc((void**)&m->range, 32);                    // words 0..3 as documented; words 4, 5: overflow in the output of the layer that carries the fused tail / in that tail's intermediate
    if (e == hipSuccess) e = hipMemset(m->range, 0, 32);
    if (e != hipSuccess) {
      set_error("eyoc_model_create: range-guard allocation failed: %s", hipGetErrorString(e));
      if (m->range) (void)hipFree(m->range);
      delete m;
      return EYOC_ERR_HIP;
    }
  }
  *out = m;
  return EYOC_OK;
}

int eyoc_model_destroy(eyoc_model* m) {
  if (!m) return EYOC_OK;
  for (auto e : m->events) (void)hipEventDestroy(e);
  if (m->range) (void)hipFree(m->range);
  delete m;
  return EYOC_OK;
}

int eyoc_model_fuse_tail(eyoc_ctx* ctx, int on) {
  if (!ctx) return -1;
  const int prev = ctx->knobs.fuse_tail;
  if (on >= 0 && on <= 2) ctx->knobs.fuse_tail = on;
  return prev;
}

int eyoc_model_set_probe(eyoc_model* m, int on) {
  EYOC_REQUIRE(m, EYOC_ERR_INVALID, "eyoc_model_set_probe: NULL model");
  const unsigned int words[2] = {0u, on ? 1u : 0u};                   // max |x| reset, probe switch
  EYOC_CHECK_HIP(hipMemcpy(m->range + 1, words, 8, hipMemcpyHostToDevice));
  m->probe = on ? 1 : 0;
  return EYOC_OK;
}

int eyoc_model_range_check(eyoc_model* m, void* stream, float* max_abs) {
  EYOC_REQUIRE(m, EYOC_ERR_INVALID, "eyoc_model_range_check: NULL model");
  unsigned int words[4] = {0, 0, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  EYOC_CHECK_HIP(hipMemcpyAsync(words, m->range, 16, hipMemcpyDeviceToHost, st));
  EYOC_CHECK_HIP(hipStreamSynchronize(st));
  float mx;
  memcpy(&mx, &words[1], 4);
  if (max_abs) *max_abs = m->probe ? mx : -1.0f;
  if (words[3]) {
    EYOC_CHECK_HIP(hipMemsetAsync(m->range + 3, 0, 4, st));            // reported once; word 0 is per forward (cleared when one starts)
    set_error("split16 arithmetic overflowed: an activation reached %g (fp16 hi halves end at 65504) in a forward since the last "
              "check - the features of every such forward are NaN; run the model with spconv math \"fp32\"", (double)SPLIT16_LIMIT);
    return EYOC_ERR_RANGE;
  }
  return EYOC_OK;
}

int eyoc_model_range_snapshot(eyoc_model* m, uint32_t* words_host, void* stream) {
  EYOC_REQUIRE(m && words_host, EYOC_ERR_INVALID, "eyoc_model_range_snapshot: NULL argument");
  EYOC_CHECK_HIP(hipMemcpyAsync(words_host, m->range, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return EYOC_OK;
}

// split-K scratch of the staged kernel (spconv_st.hip, launch_spconv_st): only stride-1 layers with >= 4 input blocks whose
// 32-channel workgroups fit KS_MAX_SLOTS partial-sum slots can split - a batch whose smallest such layer is larger (the bench: 571
// tiles x 8 channel groups at level 3) and a model that is pinned to fp32 products need none of the 32 MiB
static size_t ks_part_bytes(const eyoc_model* m, const eyoc_maps* maps) {
  if (m->math == 0) return 0;
  for (const LayerPlan& p : m->layers)
    if (p.map == M_S1 && p.cin >= 128 && p.cout >= 64 && p.cout <= 256 &&
        (long long)cdiv(maps->rows[p.level], ST_TILE) * (p.cout / 32) <= KS_MAX_SLOTS)
      return KS_PART_BYTES;
  return 0;
}

size_t eyoc_model_workspace_bytes(const eyoc_model* m, const eyoc_maps* maps) {
  if (!m || !maps) return 0;
  size_t b = 0;
  for (int i = B_X1; i < B_OUT; ++i) b += align_up((size_t)maps->rows[m->bufs[i].level] * m->bufs[i].width * sizeof(float));
  b += align_up(ks_part_bytes(m, maps));
  return b + 256;
}

int eyoc_model_num_layers(const eyoc_model* m) { return m ? (int)m->layers.size() : 0; }

int eyoc_model_set_math(eyoc_model* m, int mode) {
  EYOC_REQUIRE(m && mode >= -1 && mode <= 1, EYOC_ERR_INVALID, "eyoc_model_set_math: mode %d not in {-1, 0, 1}", mode);
  const int prev = m->math;
  m->math = mode;
  return prev + 1 ? prev + 2 : 1;   // previous mode + 2 (so that every valid answer is positive): 1 = auto, 2 = fp32, 3 = split16
}

int eyoc_model_last_math(const eyoc_model* m) { return m ? m->last_math : -1; }

int eyoc_model_set_timing(eyoc_model* m, int on) {
  EYOC_REQUIRE(m, EYOC_ERR_INVALID, "eyoc_model_set_timing: NULL model");
  m->timing = on ? 1 : 0;
  m->events_valid[0] = m->events_valid[1] = 0;
  if (on && m->events.empty()) {
    m->events.resize(2 * (m->layers.size() + 1));
    for (auto& e : m->events) EYOC_CHECK_HIP(hipEventCreate(&e));
  }
  return EYOC_OK;
}

int eyoc_model_timing_slot(eyoc_model* m, int slot) {
  EYOC_REQUIRE(m && (slot == 0 || slot == 1), EYOC_ERR_INVALID, "eyoc_model_timing_slot: slot %d", slot);
  m->slot = slot;
  return EYOC_OK;
}

int eyoc_model_layer_ms(eyoc_model* m, float* ms) {
  EYOC_REQUIRE(m && ms, EYOC_ERR_INVALID, "eyoc_model_layer_ms: NULL argument");
  EYOC_REQUIRE(m->timing && m->events_valid[m->slot], EYOC_ERR_INVALID, "eyoc_model_layer_ms: no timed forward recorded in slot %d", m->slot);
  const hipEvent_t* ev = m->events.data() + (size_t)m->slot * (m->layers.size() + 1);
  EYOC_CHECK_HIP(hipEventSynchronize(ev[m->layers.size()]));
  for (size_t i = 0; i < m->layers.size(); ++i) EYOC_CHECK_HIP(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  return EYOC_OK;
}

int eyoc_model_forward(eyoc_ctx* ctx, const eyoc_model* mc, const eyoc_maps* maps, const float* feats_dev, float* out_dev,
                       void* ws, size_t ws_bytes, void* stream) {
  EYOC_REQUIRE(ctx && mc && maps && feats_dev && out_dev && ws, EYOC_ERR_INVALID, "eyoc_model_forward: NULL argument");
  EYOC_REQUIRE(ws_bytes >= eyoc_model_workspace_bytes(mc, maps), EYOC_ERR_WORKSPACE,
               "eyoc_model_forward: workspace %zu < required %zu bytes", ws_bytes, eyoc_model_workspace_bytes(mc, maps));
  EYOC_REQUIRE(((uintptr_t)ws & 255) == 0, EYOC_ERR_INVALID, "eyoc_model_forward: workspace must be 256-byte aligned");
  eyoc_model* m = const_cast<eyoc_model*>(mc);
  hipStream_t st = (hipStream_t)stream;
  float* buf[B_COUNT];
  Carver cv(ws, ws_bytes);
  buf[B_IN] = const_cast<float*>(feats_dev);
  buf[B_OUT] = out_dev;
  for (int i = B_X1; i < B_OUT; ++i) buf[i] = cv.take<float>((size_t)maps->rows[m->bufs[i].level] * m->bufs[i].width);
  const size_t ks_bytes = ks_part_bytes(m, maps);
  float* ks_part = ks_bytes ? cv.take<float>(ks_bytes / 4) : nullptr;
  // arithmetic of the sparse convolutions: SPLIT16 needs the wave-private kernel for EVERY layer (only it reads and
  // writes the format), which round 1 measured to pay off once the finest level alone fills the chip (>= 4096 wave tiles ~ 4 KITTI
  // pairs); small batches stay on fp32, where the launcher picks the workgroup-tiled kernel per layer
  const int want = m->math;
  const bool split_ok = ctx->knobs.spconv_kernel != 0 && !(m->desc.normalize_feature && m->desc.out_channels > 64) &&
                        m->desc.channels[1] % 32 == 0;
  const bool split = split_ok && (want == 1 || (want < 0 && maps->rows[0] >= 8192));   // = the Z-order threshold of eyoc_maps_build
  EYOC_REQUIRE(want != 1 || split, EYOC_ERR_INVALID,
               "eyoc_model_forward: split16 arithmetic is not available for this model / kernel selection");
  m->last_math = split ? 1 : 0;
  // the overflow flag of THIS forward (the sticky copy is word 3).  Cleared by every forward, fp32 ones included: after an
  // overflow switched a model to fp32 MFMAs, eyoc_model_range_snapshot must not keep reporting the old verdict for forwards
  // that cannot overflow
  EYOC_CHECK_HIP(hipMemsetAsync(m->range, 0, 4, st));
  EYOC_CHECK_HIP(hipMemsetAsync(m->range + 4, 0, 8, st));             // words 4, 5: the fused tail's own verdicts (spconv_st.hip TAILF)
  hipEvent_t* ev = m->timing ? m->events.data() + (size_t)m->slot * (m->layers.size() + 1) : nullptr;
  if (m->timing) EYOC_CHECK_HIP(hipEventRecord(ev[0], st));
  bool progress_recorded = m->progress_event == nullptr;
  for (size_t li = 0; li < m->layers.size(); ++li) {
    const LayerPlan& p = m->layers[li];
    const int n_out = maps->rows[p.out_level];
    int rc;
    if (!progress_recorded && (int)li >= m->progress_layer) {            // (>=: a layer fused into its predecessor has no iteration of its own)
      EYOC_CHECK_HIP(hipEventRecord(m->progress_event, st));
      progress_recorded = true;
    }
    if (p.map == M_AFFINE) {
      rc = launch_affine(buf[p.in_buf] + p.in_col, m->bufs[p.in_buf].width, n_out, p.cout, m->blob + p.w_off, m->blob + p.b_off,
                         buf[p.out_buf] + p.out_col, m->bufs[p.out_buf].width, split, split ? m->range : nullptr, st);
    } else if (p.map == M_CONV1) {
      Conv1Args a;
      a.coords = maps->coords[0]; a.n = n_out; a.table = maps->table[0]; a.ks = m->desc.conv1_kernel_size;
      a.in = buf[p.in_buf]; a.cin = p.cin; a.w = m->blob + p.w_off; a.bias = m->blob + p.b_off; a.cout = p.cout;
      a.out = buf[p.out_buf] + p.out_col; a.ld_out = m->bufs[p.out_buf].width;
      a.out_split = split ? 1 : 0;
      a.ctx = ctx;
      a.range = split ? m->range : nullptr;
      a.wscale = m->blob + p.s_off;
      a.in_perm = maps->row_perm;                        // Z-ordered maps: the caller's features are read through the permutation
      if (a.in_perm) {
        // ... once: a copy in internal order in a buffer nothing uses yet (an indirection per probed neighbour cost the
        // 5^3 first convolution 0.4 of its 2.0 ms on the 64-pair batch)
        int best = -1;
        size_t best_sz = 0;
        for (int i = B_X1; i < B_OUT; ++i) {
          const size_t sz = (size_t)maps->rows[m->bufs[i].level] * m->bufs[i].width;
          if (i != p.out_buf && sz > best_sz) { best = i; best_sz = sz; }
        }
        if (best >= 0 && best_sz >= (size_t)n_out * p.cin) {
          rc = launch_permute_rows(a.in, a.in_perm, n_out, p.cin, buf[best], st);
          if (rc) return rc;
          a.in = buf[best];
          a.in_perm = nullptr;
        }
      }
      a.parent = maps->parent[0]; a.children = maps->children[0]; a.s1c = maps->nbr_s1[1]; a.nc = maps->rows[1];
      a.local1 = maps->row_perm ? maps->local_s1[1] : nullptr;          // Z-ordered maps: the level-1 tile rulebooks (256-parent tiles)
      rc = conv1_walks_octree(a) ? EYOC_OK : maps_build_table0(const_cast<eyoc_maps*>(maps), st);
      if (!rc) rc = launch_conv1(a, st);
    } else if (split && ctx->knobs.fuse_tail != 0 && li + 2 == m->layers.size() && p.map == M_IDENT && p.K == 1 && p.res_buf < 0 &&
               m->layers[li + 1].map == M_IDENT && m->layers[li + 1].K == 1 && m->layers[li + 1].in_buf == p.out_buf &&
               m->layers[li + 1].res_buf < 0 && !m->layers[li + 1].relu && m->layers[li + 1].out_buf == B_OUT && p.out_col == 0 &&
               tail_fusable(p.cin, p.cout, m->layers[li + 1].cout)) {
      // conv1_tr -> ReLU -> final (+ bias) -> row normalisation in one kernel: the [N, 64] intermediate stays in registers
      const LayerPlan& q = m->layers[li + 1];
      rc = launch_tail_fused(buf[p.in_buf] + p.in_col, m->bufs[p.in_buf].width, n_out, m->blob + p.w16_off, m->blob + p.s_off,
                             m->blob + p.b_off, p.relu, m->blob + q.w16_off, m->blob + q.s_off, m->blob + q.b_off, q.l2norm,
                             buf[q.out_buf] + q.out_col, m->bufs[q.out_buf].width, maps->row_perm, m->range, st);
      if (rc) return rc;
      if (m->timing) {                                  // the pair's time is booked on the first layer, the second reads 0
        EYOC_CHECK_HIP(hipEventRecord(ev[li + 1], st));
        EYOC_CHECK_HIP(hipEventRecord(ev[li + 2], st));
      }
      ++li;
      continue;
    } else {
      SpconvArgs a;
      a.nbr = p.map == M_S1 ? maps->nbr_s1[p.level] : p.map == M_DOWN ? maps->nbr_down[p.level]
              : p.map == M_UP ? maps->nbr_up[p.level] : nullptr;
      a.K = p.K; a.n_out = n_out;
      a.n_in = maps->rows[m->bufs[p.in_buf].level];
      a.in = buf[p.in_buf] + p.in_col; a.ld_in = m->bufs[p.in_buf].width; a.cin = p.cin;
      a.w = m->blob + p.w_off; a.cout = p.cout; a.bias = m->blob + p.b_off;
      a.res = p.res_buf >= 0 ? buf[p.res_buf] : nullptr; a.ld_res = p.res_buf >= 0 ? m->bufs[p.res_buf].width : 0;
      a.relu = p.relu; a.l2norm = p.l2norm;
      a.out = buf[p.out_buf] + p.out_col; a.ld_out = m->bufs[p.out_buf].width;
      a.ctx = ctx;
      if (split) {
        a.math = 1;
        a.w = m->blob + p.w16_off;
        a.out_scale = m->blob + p.s_off;
        a.out_split = p.out_buf != B_OUT;
        a.range = m->range;
        // stride-1 layers on Z-ordered rows: tile-local input stage (spconv_st.hip)
        if (p.map == M_S1 && p.cin >= 32 && p.cin % 32 == 0) { a.local = maps->local_s1[p.level]; a.ks_part = ks_part; }
        if (p.map == M_S1 && p.cin % 32 == 0 && p.cout % 128 == 0 && ctx->knobs.s1_wide) a.local128 = maps->local_s1w[p.level];
        if (p.map == M_DOWN && p.cin % 32 == 0 && p.cout % 64 == 0) a.local_down = maps->local_down[p.level];   // spconv_st.hip on 128-row tiles
        if (p.map == M_UP && p.cin % 32 == 0 && p.cout % 64 == 0) a.local_up = maps->local_up[p.level];   // spconv_up.hip
        if (p.map == M_UP && p.cin % 32 == 0 && p.cout % 64 == 0) a.local_upc = maps->local_upc[p.level]; // spconv_upc.hip (class-major tiles)
      }
      if (p.out_buf == B_OUT) a.out_perm = maps->row_perm;   // the network output goes back to the caller's row order
      // lazy tables (common.h eyoc_maps): a layer that no tile-record kernel takes reads its [27][n] table - filled here on first use
      if ((p.map == M_S1 || p.map == M_UP) && spconv_record_path(a) == 0)
        if ((rc = maps_ensure_table(const_cast<eyoc_maps*>(maps), p.map == M_S1 ? EYOC_MAP_S1 : EYOC_MAP_UP, p.level, st))) return rc;
      a.perm = p.map == M_UP ? maps->perm_up[p.level] : p.map == M_S1 ? maps->perm_s1[p.level]
               : p.map == M_DOWN ? maps->perm_down[p.level] : nullptr;
      // the 1x1 tail in the epilogue of the last staged layer (eyoc_model_fuse_tail 2, the default): block2_tr.conv2 writes the 64
      // decoder channels the tail reads next to the 32 skip channels - with the tail riding in its epilogue they never reach memory
      if (split && ctx->knobs.fuse_tail == 2 && li + 3 == m->layers.size() && p.map == M_S1 && p.cout == 64) {
        const LayerPlan &q1 = m->layers[li + 1], &q2 = m->layers[li + 2];
        if (q1.map == M_IDENT && q1.K == 1 && q1.res_buf < 0 && q2.map == M_IDENT && q2.K == 1 && q2.in_buf == q1.out_buf && q2.res_buf < 0 &&
            !q2.relu && q2.out_buf == B_OUT && q1.out_col == 0 && tail_fusable(q1.cin, q1.cout, q2.cout) && q1.in_buf == p.out_buf &&
            q1.in_col == p.out_col && q1.cin == p.cout + 32 && spconv_record_path(a) == 1 && spconv_st_can_fuse_tail(a)) {
          a.tail.skip = buf[q1.in_buf] + q1.in_col + p.cout; a.tail.ld_skip = m->bufs[q1.in_buf].width;
          a.tail.w1 = m->blob + q1.w16_off; a.tail.s1 = m->blob + q1.s_off; a.tail.b1 = m->blob + q1.b_off; a.tail.relu1 = q1.relu;
          a.tail.w2 = m->blob + q2.w16_off; a.tail.s2 = m->blob + q2.s_off; a.tail.b2 = m->blob + q2.b_off; a.tail.l2norm = q2.l2norm;
          a.tail.out = buf[q2.out_buf] + q2.out_col; a.tail.ld_out = m->bufs[q2.out_buf].width; a.tail.out_perm = maps->row_perm;
          rc = launch_spconv(a, st);
          if (rc) return rc;
          if (m->timing)                                                  // the three layers' time is booked on the first one
            for (int e = 1; e <= 3; ++e) EYOC_CHECK_HIP(hipEventRecord(ev[li + e], st));
          li += 2;
          continue;
        }
      }
      rc = launch_spconv(a, st);
    }
    if (rc) return rc;
    if (m->timing) EYOC_CHECK_HIP(hipEventRecord(ev[li + 1], st));
  }
  if (!progress_recorded) EYOC_CHECK_HIP(hipEventRecord(m->progress_event, st));
  if (m->timing) m->events_valid[m->slot] = 1;
  return EYOC_OK;
}

int eyoc_model_set_progress_event(eyoc_model* m, int layer, void* hip_event) {
  EYOC_REQUIRE(m, EYOC_ERR_INVALID, "eyoc_model_set_progress_event: NULL model");
  const int n = (int)m->layers.size();
  if (layer < 0) layer += n;                                             // -1 = in front of the last layer
  EYOC_REQUIRE(!hip_event || (layer >= 0 && layer <= n), EYOC_ERR_INVALID, "eyoc_model_set_progress_event: layer %d of %d", layer, n);
  m->progress_event = (hipEvent_t)hip_event;
  m->progress_layer = layer;
  return EYOC_OK;
}

int eyoc_model_layer_work(eyoc_ctx* ctx, const eyoc_model* m, const eyoc_maps* maps, void* stream, const char** names,
                          int64_t* pairs, double* flops, double* gather_bytes, double* compulsory_bytes) {
  EYOC_REQUIRE(ctx && m && maps, EYOC_ERR_INVALID, "eyoc_model_layer_work: NULL argument");
  eyoc_maps_info_t info;
  int rc = eyoc_maps_info(ctx, maps, m->desc.conv1_kernel_size, stream, &info);
  if (rc) return rc;
  for (size_t li = 0; li < m->layers.size(); ++li) {
    const LayerPlan& p = m->layers[li];
    const double n_out = maps->rows[p.out_level];
    double n_in = n_out, pr = 0;
    switch (p.map) {
      case M_CONV1: pr = (double)info.pairs_conv1; break;
      case M_S1: pr = (double)info.pairs_s1[p.level]; break;
      case M_DOWN: pr = (double)info.pairs_down[p.level]; n_in = maps->rows[p.level]; break;
      case M_UP: pr = (double)info.pairs_up[p.level]; n_in = maps->rows[p.level + 1]; break;
      case M_IDENT: pr = n_out; break;
      case M_AFFINE: break;
    }
    if (p.map == M_AFFINE) {                                             // no products: one read and one write of the rows
      if (names) names[li] = p.name.c_str();
      if (pairs) pairs[li] = 0;
      if (flops) flops[li] = 2.0 * n_out * p.cout;
      if (gather_bytes) gather_bytes[li] = 8.0 * n_out * p.cout;
      if (compulsory_bytes) compulsory_bytes[li] = 8.0 * n_out * p.cout;
      continue;
    }
    const double wbytes = 4.0 * p.K * p.cin * p.cout;
    if (names) names[li] = p.name.c_str();
    if (pairs) pairs[li] = (int64_t)pr;
    if (flops) flops[li] = 2.0 * pr * p.cin * p.cout;
    if (gather_bytes) gather_bytes[li] = pr * (4.0 * p.cin + 8.0) + 4.0 * n_out * p.cout + wbytes;
    if (compulsory_bytes) compulsory_bytes[li] = 4.0 * (n_in * p.cin + n_out * p.cout) + 8.0 * pr + wbytes;
  }
  return EYOC_OK;
}

}  // extern "C"
