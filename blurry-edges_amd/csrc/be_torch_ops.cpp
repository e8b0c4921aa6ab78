// torch.ops.be.*: the C entry points of libblurry_edges_hip registered as PyTorch operators (north_star: "hand-written HIP C++
// kernels ... exposed as torch extensions").  This file is host-only glue over the C ABI of include/blurry_edges_hip.h - it
// allocates outputs with torch, takes the current stream from torch and calls the same symbols the ctypes binding
// (be_hip/native.py) calls; the C ABI stays the language-neutral boundary.  Registered for the LocalStage hot path: the packed
// inference forward (models/local_stage.py:63-73), pass-A colours and the depth solve (bench.py's step), and the training
// units, pooling, loss and optimizer tail of the training step (local_training.py:103-108).
// Host structs of the C ABI (be_render_opts, be_depth_consts, the device-side job tables) travel as byte tensors.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/blurry_edges_hip.h"

namespace {

using at::Tensor;
using c10::optional;

void* stream_of(const Tensor& t) {
    // arguments are evaluated in any order: refuse a CPU tensor here too, before its device index reaches the HIP runtime
    TORCH_CHECK(t.is_cuda(), "expected a tensor on the GPU; the HIP path has no CPU fallback");
    return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

const float* fp(const Tensor& t, const char* what) {
    TORCH_CHECK(t.is_cuda(), what, ": expected a tensor on the GPU; the HIP path has no CPU fallback");
    TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), what, ": expected a contiguous float32 tensor");
    return t.data_ptr<float>();
}
float* fpm(const Tensor& t, const char* what) { return const_cast<float*>(fp(t, what)); }
const float* fpo(const optional<Tensor>& t, const char* what) { return t.has_value() && t->defined() ? fp(*t, what) : nullptr; }

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", be_last_error()); }

template <class T>
const T* host_struct(const Tensor& bytes, const char* what) {
    TORCH_CHECK(!bytes.is_cuda() && bytes.scalar_type() == at::kByte && bytes.is_contiguous() && (size_t)bytes.numel() >= sizeof(T), what,
                ": expected a CPU uint8 tensor holding the C struct");
    return reinterpret_cast<const T*>(bytes.data_ptr<uint8_t>());
}

// ---- inference -----------------------------------------------------------------------------------------------------------
Tensor local_stage_pack(at::TensorList tensors, double eps) {
    TORCH_CHECK(tensors.size() == 86, "local_stage_pack: 86 tensors expected (native.local_stage_pack order)");
    std::vector<const float*> ptrs;
    std::vector<Tensor> keep;
    for (const Tensor& t : tensors) { keep.push_back(t.contiguous()); ptrs.push_back(fp(keep.back(), "local_stage_pack")); }
    Tensor packed = at::empty({(int64_t)be_local_stage_packed_floats()}, keep[0].options());
    check(be_local_stage_pack_f32(ptrs.data(), (float)eps, packed.data_ptr<float>(), stream_of(packed)), "be_local_stage_pack_f32");
    return packed;
}

std::tuple<Tensor, Tensor> local_stage_forward(const Tensor& packed, const Tensor& x, const optional<Tensor>& out_,
                                               const optional<Tensor>& workspace_, bool winograd, int64_t chunk) {
    TORCH_CHECK(x.dim() == 4 && x.size(1) == 3 && x.size(2) == BE_R && x.size(3) == BE_R, "local_stage_forward: x must be [N,3,21,21]");
    const int64_t n = x.size(0);
    Tensor out = out_.has_value() && out_->defined() ? *out_ : at::empty({n, 10}, x.options());
    const size_t need = be_local_stage_workspace_bytes(n, (int64_t)chunk);
    Tensor ws = workspace_.has_value() && workspace_->defined() && (size_t)workspace_->numel() * 4 >= need
                    ? *workspace_ : at::empty({(int64_t)((need + 3) / 4)}, x.options());
    be_local_stage_opts o{winograd ? 1 : 0, (int)chunk};
    check(be_local_stage_forward_f32(fp(packed, "packed"), fp(x, "x"), fpm(out, "out"), n, ws.data_ptr<float>(), (size_t)ws.numel() * 4, &o,
                                     stream_of(x)), "be_local_stage_forward_f32");
    return {out, ws};
}

Tensor render_colors(const Tensor& opts, const Tensor& params10, const Tensor& patches, const optional<Tensor>& colors_) {
    TORCH_CHECK(params10.is_cuda() && patches.is_cuda(), "render_colors: expected tensors on the GPU; the HIP path has no CPU fallback");
    const int64_t n = params10.size(0);
    TORCH_CHECK(params10.dim() == 2 && params10.size(1) == 10 && patches.numel() == n * 3 * BE_NPIX, "render_colors: params10 [N,10], patches [N,3,21,21]");
    Tensor colors = colors_.has_value() && colors_->defined() ? *colors_ : at::empty({n, 3, 3}, params10.options());
    check(be_render_colors_f32(host_struct<be_render_opts>(opts, "render_colors(opts)"), fp(params10, "params10"), fp(patches, "patches"),
                               fpm(colors, "colors"), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n, stream_of(params10)),
          "be_render_colors_f32");
    return colors;
}

Tensor local_depth(const Tensor& consts, const Tensor& params10, const optional<Tensor>& out_) {
    TORCH_CHECK(params10.is_cuda(), "local_depth: expected a tensor on the GPU; the HIP path has no CPU fallback");
    TORCH_CHECK(params10.dim() == 2 && params10.size(1) == 10 && params10.size(0) % 2 == 0, "local_depth: params10 [2P,10]");
    const int64_t p = params10.size(0) / 2;
    Tensor out = out_.has_value() && out_->defined() ? *out_ : at::empty({p, 2}, params10.options());
    check(be_local_depth_f32(host_struct<be_depth_consts>(consts, "local_depth(consts)"), fp(params10, "params10"), fpm(out, "out"), p,
                             stream_of(params10)), "be_local_depth_f32");
    return out;
}

// ---- training ------------------------------------------------------------------------------------------------------------
// conv / linear + BatchNorm (batch statistics) [+ res] [+ Smish]: -> (out, y, mean, invstd, s_in or an empty tensor)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> train_unit_fwd(const Tensor& x, const Tensor& pw, const Tensor& pb, const Tensor& gamma,
                                                                  const Tensor& beta, const optional<Tensor>& res, Tensor run_mean,
                                                                  Tensor run_var, int64_t cout, int64_t ksize, bool act, double eps,
                                                                  double momentum, Tensor scratch) {
    TORCH_CHECK(x.dim() == 4, "train_unit_fwd: x must be NHWC [N,H,W,C]");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3);
    Tensor y = at::empty({n, h, w, cout}, x.options()), out = at::empty_like(y);
    Tensor s_in = act ? at::empty_like(y) : at::empty({0}, x.options());
    Tensor mean = at::empty({cout}, x.options()), invstd = at::empty({cout}, x.options());
    be_conv_desc d{(int)n, (int)h, (int)w, (int)cin, (int)cout, (int)ksize, 0};
    check(be_train_unit_fwd_f32(&d, fp(x, "x"), fp(pw, "packed_w"), fp(pb, "packed_bias"), fp(gamma, "gamma"), fp(beta, "beta"), fpo(res, "res"),
                                (float)eps, (float)momentum, fpm(run_mean, "run_mean"), fpm(run_var, "run_var"), y.data_ptr<float>(),
                                mean.data_ptr<float>(), invstd.data_ptr<float>(), act ? s_in.data_ptr<float>() : nullptr,
                                out.data_ptr<float>(), act ? 1 : 0, scratch.data_ptr<float>(), (size_t)scratch.numel() * 4, stream_of(x)),
          "be_train_unit_fwd_f32");
    return {out, y, mean, invstd, s_in};
}

// backward of the unit; the four parameter gradients are written into the given slices of the flat gradient buffer
std::tuple<Tensor, Tensor> train_unit_bwd(const Tensor& x, const Tensor& dout, const optional<Tensor>& s_in, const Tensor& y, const Tensor& mean,
                                          const Tensor& invstd, const Tensor& gamma, const optional<Tensor>& dg_pw,
                                          const optional<Tensor>& dg_pb, const optional<Tensor>& dx_add, int64_t ksize, int64_t chw_hw,
                                          Tensor dgamma, Tensor dbeta, Tensor dw, Tensor db, Tensor scratch) {
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3), cout = y.size(-1);
    Tensor ds = at::empty_like(y), dy = at::empty_like(y);
    const bool want_dx = dg_pw.has_value() && dg_pw->defined();
    Tensor dx = want_dx ? at::empty({n, h, w, cin}, x.options()) : at::empty({0}, x.options());
    be_conv_desc d{(int)n, (int)h, (int)w, (int)cin, (int)cout, (int)ksize, 0};
    const bool has_s = s_in.has_value() && s_in->defined() && s_in->numel() > 0;
    check(be_train_unit_bwd_f32(&d, fp(x, "x"), fp(dout, "dout"), has_s ? fp(*s_in, "s_in") : nullptr, fp(y, "y"), fp(mean, "mean"),
                                fp(invstd, "invstd"), fp(gamma, "gamma"), want_dx ? fp(*dg_pw, "dgrad_w") : nullptr,
                                want_dx ? fpo(dg_pb, "dgrad_b") : nullptr, fpo(dx_add, "dx_add"), (int)chw_hw, ds.data_ptr<float>(),
                                dy.data_ptr<float>(), fpm(dgamma, "dgamma"), fpm(dbeta, "dbeta"), fpm(dw, "dw"), fpm(db, "db"),
                                want_dx ? dx.data_ptr<float>() : nullptr, scratch.data_ptr<float>(), (size_t)scratch.numel() * 4, stream_of(x)),
          "be_train_unit_bwd_f32");
    return {ds, dx};
}

// two units on one input with shared launches (a residual block's 3x3 convolution and 1x1 downsample): the outputs of two
// train_unit_fwd calls, a's five then b's five
std::vector<Tensor> train_unit_pair_fwd(const Tensor& x, const Tensor& pw_a, const Tensor& pb_a, const Tensor& gamma_a, const Tensor& beta_a,
                                        Tensor rm_a, Tensor rv_a, int64_t cout_a, int64_t ks_a, bool act_a, const Tensor& pw_b,
                                        const Tensor& pb_b, const Tensor& gamma_b, const Tensor& beta_b, Tensor rm_b, Tensor rv_b,
                                        int64_t cout_b, int64_t ks_b, bool act_b, double eps, double momentum, Tensor scratch) {
    TORCH_CHECK(x.dim() == 4, "train_unit_pair_fwd: x must be NHWC [N,H,W,C]");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3);
    std::vector<Tensor> r;
    be_train_unit_fwd u[2];
    const Tensor* pw[2] = {&pw_a, &pw_b}; const Tensor* pb[2] = {&pb_a, &pb_b};
    const Tensor* ga[2] = {&gamma_a, &gamma_b}; const Tensor* be[2] = {&beta_a, &beta_b};
    Tensor* rm[2] = {&rm_a, &rm_b}; Tensor* rv[2] = {&rv_a, &rv_b};
    const int64_t cout[2] = {cout_a, cout_b}, ks[2] = {ks_a, ks_b};
    const bool act[2] = {act_a, act_b};
    for (int j = 0; j < 2; ++j) {
        Tensor y = at::empty({n, h, w, cout[j]}, x.options()), out = at::empty_like(y);
        Tensor s_in = act[j] ? at::empty_like(y) : at::empty({0}, x.options());
        Tensor mean = at::empty({cout[j]}, x.options()), invstd = at::empty({cout[j]}, x.options());
        u[j] = be_train_unit_fwd{be_conv_desc{(int)n, (int)h, (int)w, (int)cin, (int)cout[j], (int)ks[j], 0}, fp(x, "x"), fp(*pw[j], "packed_w"),
                                 fp(*pb[j], "packed_bias"), fp(*ga[j], "gamma"), fp(*be[j], "beta"), nullptr, fpm(*rm[j], "run_mean"),
                                 fpm(*rv[j], "run_var"), y.data_ptr<float>(), mean.data_ptr<float>(), invstd.data_ptr<float>(),
                                 act[j] ? s_in.data_ptr<float>() : nullptr, out.data_ptr<float>(), act[j] ? 1 : 0};
        r.insert(r.end(), {out, y, mean, invstd, s_in});
    }
    check(be_train_unit_pair_fwd_f32(&u[0], &u[1], (float)eps, (float)momentum, scratch.data_ptr<float>(), (size_t)scratch.numel() * 4,
                                     stream_of(x)), "be_train_unit_pair_fwd_f32");
    return r;
}

// backward of the pair: -> (ds_a, ds_b, dx) with dx = the sum of the two units' input gradients
std::vector<Tensor> train_unit_pair_bwd(const Tensor& x, const Tensor& dout_a, const optional<Tensor>& s_in_a, const Tensor& y_a,
                                        const Tensor& mean_a, const Tensor& invstd_a, const Tensor& gamma_a, const Tensor& dg_pw_a,
                                        const Tensor& dg_pb_a, int64_t ks_a, Tensor dgamma_a, Tensor dbeta_a, Tensor dw_a, Tensor db_a,
                                        const Tensor& dout_b, const optional<Tensor>& s_in_b, const Tensor& y_b, const Tensor& mean_b,
                                        const Tensor& invstd_b, const Tensor& gamma_b, const Tensor& dg_pw_b, const Tensor& dg_pb_b,
                                        int64_t ks_b, Tensor dgamma_b, Tensor dbeta_b, Tensor dw_b, Tensor db_b, Tensor scratch) {
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3);
    const Tensor* dout[2] = {&dout_a, &dout_b}; const optional<Tensor>* s_in[2] = {&s_in_a, &s_in_b};
    const Tensor* y[2] = {&y_a, &y_b}; const Tensor* mean[2] = {&mean_a, &mean_b}; const Tensor* invstd[2] = {&invstd_a, &invstd_b};
    const Tensor* gamma[2] = {&gamma_a, &gamma_b}; const Tensor* pw[2] = {&dg_pw_a, &dg_pw_b}; const Tensor* pb[2] = {&dg_pb_a, &dg_pb_b};
    Tensor* dgamma[2] = {&dgamma_a, &dgamma_b}; Tensor* dbeta[2] = {&dbeta_a, &dbeta_b}; Tensor* dw[2] = {&dw_a, &dw_b};
    Tensor* db[2] = {&db_a, &db_b};
    const int64_t ks[2] = {ks_a, ks_b};
    be_train_unit_bwd u[2];
    std::vector<Tensor> ds, dy, dx;
    for (int j = 0; j < 2; ++j) {
        ds.push_back(at::empty_like(*y[j])); dy.push_back(at::empty_like(*y[j])); dx.push_back(at::empty({n, h, w, cin}, x.options()));
        const bool has_s = s_in[j]->has_value() && (*s_in[j])->defined() && (*s_in[j])->numel() > 0;
        u[j] = be_train_unit_bwd{be_conv_desc{(int)n, (int)h, (int)w, (int)cin, (int)y[j]->size(-1), (int)ks[j], 0}, fp(x, "x"), fp(*dout[j], "dout"),
                                 has_s ? fp(**s_in[j], "s_in") : nullptr, fp(*y[j], "y"), fp(*mean[j], "mean"), fp(*invstd[j], "invstd"),
                                 fp(*gamma[j], "gamma"), fp(*pw[j], "dgrad_w"), fp(*pb[j], "dgrad_b"), nullptr, 0, ds[j].data_ptr<float>(),
                                 dy[j].data_ptr<float>(), fpm(*dgamma[j], "dgamma"), fpm(*dbeta[j], "dbeta"), fpm(*dw[j], "dw"), fpm(*db[j], "db"),
                                 dx[j].data_ptr<float>()};
    }
    check(be_train_unit_pair_bwd_f32(&u[0], &u[1], scratch.data_ptr<float>(), (size_t)scratch.numel() * 4, stream_of(x)),
          "be_train_unit_pair_bwd_f32");
    return {ds[0], ds[1], dx[0]};
}

std::tuple<Tensor, Tensor> maxpool_fwd_idx(const Tensor& x, int64_t k, int64_t stride, int64_t pad) {
    TORCH_CHECK(x.dim() == 4, "maxpool_fwd_idx: x must be NHWC [N,H,W,C]");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), c = x.size(3);
    const int64_t oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    Tensor y = at::empty({n, oh, ow, c}, x.options()), idx = at::empty({n, oh, ow, c}, x.options().dtype(at::kByte));
    check(be_maxpool_nhwc_fwd_idx_f32(fp(x, "x"), y.data_ptr<float>(), idx.data_ptr<uint8_t>(), (int)n, (int)h, (int)w, (int)c, (int)k, (int)stride,
                                      (int)pad, stream_of(x)), "be_maxpool_nhwc_fwd_idx_f32");
    return {y, idx};
}

Tensor maxpool_bwd_idx(const Tensor& idx, const Tensor& dout, int64_t h, int64_t w, int64_t k, int64_t stride, int64_t pad) {
    TORCH_CHECK(dout.dim() == 4 && idx.is_cuda() && idx.scalar_type() == at::kByte && idx.is_contiguous() && idx.numel() == dout.numel(),
                "maxpool_bwd_idx: dout [N,oh,ow,C] float32 and idx = the uint8 winners of maxpool_fwd_idx (same shape) on the GPU");
    const int64_t n = dout.size(0), c = dout.size(3);
    Tensor dx = at::empty({n, h, w, c}, dout.options());
    check(be_maxpool_nhwc_bwd_idx_f32(idx.data_ptr<uint8_t>(), fp(dout, "dout"), dx.data_ptr<float>(), (int)n, (int)h, (int)w, (int)c, (int)k,
                                      (int)stride, (int)pad, stream_of(dout)), "be_maxpool_nhwc_bwd_idx_f32");
    return dx;
}

// clip_grad_norm_(max_norm) + AdamW over the flat gradient buffer; returns nothing (grad_norm is written in place)
void clip_adamw(const Tensor& table, int64_t nentries, Tensor grad_flat, Tensor partial, double max_norm, double grad_scale, double lr,
                double beta1, double beta2, double eps, double weight_decay, Tensor step, Tensor grad_norm, bool write_back) {
    TORCH_CHECK(table.is_cuda() && table.scalar_type() == at::kByte && partial.scalar_type() == at::kDouble, "clip_adamw: table (uint8) / partial (float64)");
    check(be_clip_adamw_f32(reinterpret_cast<const be_adam_entry*>(table.data_ptr<uint8_t>()), (int)nentries, fpm(grad_flat, "grad_flat"),
                            grad_flat.numel(), partial.data_ptr<double>(), (int)partial.numel(), (float)max_norm, (float)grad_scale, lr, beta1,
                            beta2, eps, weight_decay, fpm(step, "step"), fpm(grad_norm, "grad_norm"), write_back ? 1 : 0, stream_of(grad_flat)),
          "be_clip_adamw_f32");
}


// ---- the fine-grained PostProcess / DepthEtas methods and their adjoints (be_compat.hip, be_compat_bwd.hip, be_elementwise.hip) ------
// flat one-row-per-patch layouts; the Python classes permute the reference's [B,K,...,Hp,Wp] layout to and from them
Tensor params2etas(const Tensor& p) {
    Tensor out = at::empty_like(p);
    check(be_params2etas_f32(fp(p, "params"), out.data_ptr<float>(), p.numel(), stream_of(p)), "be_params2etas_f32");
    return out;
}
Tensor params2etas_bwd(const Tensor& p, const Tensor& geta) {
    TORCH_CHECK(geta.numel() == p.numel(), "params2etas_bwd: cotangent size");
    Tensor gp = at::empty_like(p);
    check(be_params2etas_bwd_f32(fp(p, "params"), fp(geta, "geta"), gp.data_ptr<float>(), p.numel(), stream_of(p)), "be_params2etas_bwd_f32");
    return gp;
}
Tensor params2dists(const Tensor& opts, const Tensor& params8) {
    TORCH_CHECK(params8.dim() == 2 && params8.size(1) == 8, "params2dists: params8 [N,8]");
    const int64_t n = params8.size(0);
    fp(params8, "params8");
    Tensor out = at::empty({n, 2, BE_R, BE_R}, params8.options());
    check(be_params2dists_f32(host_struct<be_render_opts>(opts, "params2dists(opts)"), params8.data_ptr<float>(), out.data_ptr<float>(), n,
                              stream_of(params8)), "be_params2dists_f32");
    return out;
}
Tensor params2dists_bwd(const Tensor& opts, const Tensor& params8, const Tensor& gdists) {
    TORCH_CHECK(params8.dim() == 2 && params8.size(1) == 8, "params2dists_bwd: params8 [N,8]");
    const int64_t n = params8.size(0);
    TORCH_CHECK(gdists.numel() == n * 2 * BE_NPIX, "params2dists_bwd: gdists [N,2,21,21]");
    fp(params8, "params8");
    Tensor g = at::empty_like(params8);
    check(be_params2dists_bwd_f32(host_struct<be_render_opts>(opts, "params2dists_bwd(opts)"), params8.data_ptr<float>(), fp(gdists, "gdists"),
                                  g.data_ptr<float>(), n, stream_of(params8)), "be_params2dists_bwd_f32");
    return g;
}
Tensor dists2indicators(const Tensor& dists, const Tensor& etas) {
    const int64_t n = etas.numel() / 2;
    TORCH_CHECK(dists.numel() == n * 2 * BE_NPIX && etas.numel() == 2 * n, "dists2indicators: dists [N,2,21,21], etas [N,2]");
    Tensor out = at::empty({n, 3, BE_R, BE_R}, dists.options());
    check(be_dists2indicators_f32(fp(dists, "dists"), fp(etas, "etas"), out.data_ptr<float>(), n, stream_of(dists)), "be_dists2indicators_f32");
    return out;
}
std::tuple<Tensor, Tensor> dists2indicators_bwd(const Tensor& dists, const Tensor& etas, const Tensor& gw) {
    const int64_t n = etas.numel() / 2;
    TORCH_CHECK(dists.numel() == n * 2 * BE_NPIX && gw.numel() == n * 3 * BE_NPIX, "dists2indicators_bwd: dists [N,2,21,21], gwedges [N,3,21,21]");
    Tensor gd = at::empty_like(dists), ge = at::empty_like(etas);
    check(be_dists2indicators_bwd_f32(fp(dists, "dists"), fp(etas, "etas"), fp(gw, "gwedges"), gd.data_ptr<float>(), ge.data_ptr<float>(), n,
                                      stream_of(dists)), "be_dists2indicators_bwd_f32");
    return {gd, ge};
}
Tensor inverse3x3(const Tensor& a) {
    TORCH_CHECK(a.numel() % 9 == 0, "inverse3x3: [...,3,3]");
    Tensor out = at::empty_like(a);
    check(be_inverse3x3_f32(fp(a, "A"), out.data_ptr<float>(), a.numel() / 9, stream_of(a)), "be_inverse3x3_f32");
    return out;
}
Tensor inverse3x3_bwd(const Tensor& inv, const Tensor& gout) {
    TORCH_CHECK(inv.numel() % 9 == 0 && gout.numel() == inv.numel(), "inverse3x3_bwd: [...,3,3]");
    Tensor ga = at::empty_like(inv);
    check(be_inverse3x3_bwd_f32(fp(inv, "inv"), fp(gout, "gout"), ga.data_ptr<float>(), inv.numel() / 9, stream_of(inv)), "be_inverse3x3_bwd_f32");
    return ga;
}
Tensor image_derivative(const Tensor& img) {
    TORCH_CHECK(img.dim() == 4, "image_derivative: img [N,C,H,W]");
    const int64_t n = img.size(0), c = img.size(1), h = img.size(2), w = img.size(3);
    Tensor out = at::empty({n, c, h - 2, w - 2}, img.options());
    check(be_image_derivative_f32(fp(img, "img"), out.data_ptr<float>(), n * c, (int)h, (int)w, stream_of(img)), "be_image_derivative_f32");
    return out;
}
Tensor image_derivative_bwd(const Tensor& img, const Tensor& gout) {
    TORCH_CHECK(img.dim() == 4, "image_derivative_bwd: img [N,C,H,W]");
    const int64_t n = img.size(0), c = img.size(1), h = img.size(2), w = img.size(3);
    TORCH_CHECK(gout.numel() == n * c * (h - 2) * (w - 2), "image_derivative_bwd: gout [N,C,H-2,W-2]");
    Tensor g = at::empty_like(img);
    check(be_image_derivative_bwd_f32(fp(img, "img"), fp(gout, "gout"), g.data_ptr<float>(), n * c, (int)h, (int)w, stream_of(img)),
          "be_image_derivative_bwd_f32");
    return g;
}
Tensor normalized_gaussian(const Tensor& x, double delta_sq) {
    Tensor y = at::empty_like(x);
    check(be_normalized_gaussian_f32(fp(x, "x"), y.data_ptr<float>(), (float)delta_sq, x.numel(), stream_of(x)), "be_normalized_gaussian_f32");
    return y;
}
Tensor normalized_gaussian_bwd(const Tensor& x, const Tensor& gy, double delta_sq) {
    TORCH_CHECK(gy.numel() == x.numel(), "normalized_gaussian_bwd: cotangent size");
    Tensor gx = at::empty_like(x);
    check(be_normalized_gaussian_bwd_f32(fp(x, "x"), fp(gy, "gy"), gx.data_ptr<float>(), (float)delta_sq, x.numel(), stream_of(x)),
          "be_normalized_gaussian_bwd_f32");
    return gx;
}
Tensor etas2depth(const Tensor& consts, const Tensor& eta1, const Tensor& eta2) {
    TORCH_CHECK(eta1.numel() == eta2.numel(), "etas2depth: eta1 / eta2 sizes");
    fp(eta1, "eta1");
    Tensor z = at::empty_like(eta1);
    check(be_etas2depth_f32(host_struct<be_depth_consts>(consts, "etas2depth(consts)"), eta1.data_ptr<float>(), fp(eta2, "eta2"), z.data_ptr<float>(),
                            nullptr, eta1.numel(), stream_of(eta1)), "be_etas2depth_f32");
    return z;
}
std::tuple<Tensor, Tensor> etas2depth_bwd(const Tensor& consts, const Tensor& eta1, const Tensor& eta2, const Tensor& gz) {
    TORCH_CHECK(eta1.numel() == eta2.numel() && gz.numel() == eta1.numel(), "etas2depth_bwd: sizes");
    fp(eta1, "eta1");
    Tensor g1 = at::empty_like(eta1), g2 = at::empty_like(eta1);
    check(be_etas2depth_bwd_f32(host_struct<be_depth_consts>(consts, "etas2depth_bwd(consts)"), eta1.data_ptr<float>(), fp(eta2, "eta2"),
                                fp(gz, "gdepth"), g1.data_ptr<float>(), g2.data_ptr<float>(), eta1.numel(), stream_of(eta1)), "be_etas2depth_bwd_f32");
    return {g1, g2};
}
Tensor depth2sigma(const Tensor& consts, const Tensor& depth, double rho_prime) {
    fp(depth, "depth");
    Tensor out = at::empty_like(depth);
    check(be_depth2sigma_f32(host_struct<be_depth_consts>(consts, "depth2sigma(consts)"), depth.data_ptr<float>(), (float)rho_prime,
                             out.data_ptr<float>(), depth.numel(), stream_of(depth)), "be_depth2sigma_f32");
    return out;
}
Tensor depth2sigma_bwd(const Tensor& consts, const Tensor& depth, double rho_prime, const Tensor& geta) {
    TORCH_CHECK(geta.numel() == depth.numel(), "depth2sigma_bwd: cotangent size");
    fp(depth, "depth");
    Tensor g = at::empty_like(depth);
    check(be_depth2sigma_bwd_f32(host_struct<be_depth_consts>(consts, "depth2sigma_bwd(consts)"), depth.data_ptr<float>(), (float)rho_prime,
                                 fp(geta, "geta"), g.data_ptr<float>(), depth.numel(), stream_of(depth)), "be_depth2sigma_bwd_f32");
    return g;
}
// nn.Fold of a [B,C,21,21,Hp,Wp] patch tensor (contiguous) -> [B,C,H,W]; mode 0 sum, 1 mean, 2 count of positive entries
Tensor fold_patches(const Tensor& src, int64_t B, int64_t C, int64_t hp, int64_t wp, int64_t H, int64_t W, int64_t stride, int64_t mode) {
    TORCH_CHECK(src.is_cuda() && src.is_contiguous(), "fold_patches: expected a contiguous tensor on the GPU; the HIP path has no CPU fallback");
    const int64_t p = hp * wp;
    TORCH_CHECK(src.numel() == B * C * BE_NPIX * p, "fold_patches: src [B,C,21,21,Hp,Wp]");
    const bool is_int = src.scalar_type() == at::kInt;
    TORCH_CHECK(is_int ? mode == 2 : src.scalar_type() == at::kFloat, "fold_patches: float32 source (int32 only for mode 2)");
    Tensor out = at::empty({B, C, H, W}, src.options().dtype(at::kFloat));
    check(be_fold_patches_f32(is_int ? nullptr : src.data_ptr<float>(), is_int ? src.data_ptr<int32_t>() : nullptr, out.data_ptr<float>(), (int)B,
                              (int)C, (int)hp, (int)wp, (int)H, (int)W, (int)stride, C * BE_NPIX * p, BE_NPIX * p, BE_R * p, p, wp, 1, (int)mode,
                              stream_of(src)), "be_fold_patches_f32");
    return out;
}
Tensor fold_patches_bwd(const Tensor& gout, int64_t hp, int64_t wp, int64_t stride, int64_t mode) {
    TORCH_CHECK(gout.dim() == 4, "fold_patches_bwd: gout [B,C,H,W]");
    const int64_t B = gout.size(0), C = gout.size(1), H = gout.size(2), W = gout.size(3), p = hp * wp;
    Tensor g = at::empty({B, C, BE_R, BE_R, hp, wp}, gout.options());
    check(be_fold_patches_bwd_f32(fp(gout, "gout"), g.data_ptr<float>(), (int)B, (int)C, (int)hp, (int)wp, (int)H, (int)W, (int)stride,
                                  C * BE_NPIX * p, BE_NPIX * p, BE_R * p, p, wp, 1, (int)mode, stream_of(gout)), "be_fold_patches_bwd_f32");
    return g;
}
void wrap_angles_(Tensor est, int64_t col0, int64_t col1) {
    TORCH_CHECK(est.dim() == 2, "wrap_angles_: est [N,ld]");
    check(be_wrap_angles_inplace_f32(fpm(est, "est"), est.size(0), (int)est.size(1), (int)col0, (int)col1, stream_of(est)),
          "be_wrap_angles_inplace_f32");
}


// ---- pass B, folds, the two fused losses, attention (the rest of north_star's "exposed as torch extensions") ----------
// a be_patch_view travels as its bytes (it holds the device pointer of `pixels`, which is passed next to it so that the operator
// sees - and keeps alive - the tensor the view points into)
std::vector<Tensor> render_full(const Tensor& opts, const Tensor& consts, double rho_prime, bool densify_w, const Tensor& params12,
                                const Tensor& view, const Tensor& pixels, int64_t want) {
    TORCH_CHECK(params12.dim() == 2 && params12.size(1) == 12, "render_full: params12 [P,12]");
    TORCH_CHECK(pixels.is_cuda(), "render_full: expected the viewed pixels on the GPU; the HIP path has no CPU fallback");
    fp(params12, "params12");
    const int64_t n = params12.size(0);
    const be_patch_view* v = host_struct<be_patch_view>(view, "render_full(view)");
    auto o = params12.options();
    Tensor rec = at::empty({n, BE_RECORD_FLOATS}, o);
    Tensor t[6];
    if (want & 1) t[0] = at::empty({n, 2, 3, BE_R, BE_R}, o);
    if (want & 2) t[1] = at::empty({n, 3, BE_R, BE_R}, o);
    if (want & 4) t[2] = at::empty({n, 3, BE_R, BE_R}, o);
    if (want & 8) t[3] = at::empty({n, BE_R, BE_R}, o);
    if (want & 16) t[4] = at::empty({n, BE_R, BE_R}, o);
    if (want & 32) t[5] = at::empty({n, BE_R, BE_R}, o.dtype(at::kInt));
    auto f = [&](int i) { return t[i].defined() ? t[i].data_ptr<float>() : nullptr; };
    check(be_render_full_f32(host_struct<be_render_opts>(opts, "render_full(opts)"), host_struct<be_depth_consts>(consts, "render_full(consts)"),
                             (float)rho_prime, densify_w ? 1 : 0, params12.data_ptr<float>(), v, rec.data_ptr<float>(), f(0), f(1), f(2), f(3), f(4),
                             t[5].defined() ? t[5].data_ptr<int32_t>() : nullptr, n, stream_of(params12)), "be_render_full_f32");
    std::vector<Tensor> r{rec};
    for (int i = 0; i < 6; ++i) if (t[i].defined()) r.push_back(t[i]);
    return r;
}

// records [P,32] or [B,P,32] -> the maps named by `want` (bit 0 image, 1 shpd, 2 refoc, 3 bndry, 4 depth, 5 conf), in that order
std::vector<Tensor> fold_records(const Tensor& opts, const Tensor& records, int64_t hp, int64_t wp, int64_t H, int64_t W, int64_t stride,
                                 bool densify_w, int64_t want) {
    fp(records, "records");
    const bool batched = records.dim() == 3;
    TORCH_CHECK((batched || records.dim() == 2) && records.size(-1) == BE_RECORD_FLOATS && records.size(-2) == hp * wp,
                "fold_records: records [P,32] or [B,P,32] with P = hp * wp");
    const int64_t B = batched ? records.size(0) : 1;
    auto o = records.options();
    auto shape = [&](std::vector<int64_t> s) { if (batched) s.insert(s.begin(), B); return s; };
    Tensor t[6];
    if (want & 1) t[0] = at::empty(shape({2, 3, H, W}), o);
    if (want & 2) t[1] = at::empty(shape({3, H, W}), o);
    if (want & 4) t[2] = at::empty(shape({3, H, W}), o);
    if (want & 8) t[3] = at::empty(shape({H, W}), o);
    if (want & 16) t[4] = at::empty(shape({H, W}), o);
    if (want & 32) t[5] = at::empty(shape({H, W}), o);
    auto f = [&](int i) { return t[i].defined() ? t[i].data_ptr<float>() : nullptr; };
    const be_render_opts* ro = host_struct<be_render_opts>(opts, "fold_records(opts)");
    if (batched)
        check(be_fold_records_batch_f32(ro, records.data_ptr<float>(), (int)B, (int)hp, (int)wp, (int)H, (int)W, (int)stride, densify_w ? 1 : 0, f(0),
                                        f(1), f(2), f(3), f(4), f(5), stream_of(records)), "be_fold_records_batch_f32");
    else
        check(be_fold_records_f32(ro, records.data_ptr<float>(), (int)hp, (int)wp, (int)H, (int)W, (int)stride, densify_w ? 1 : 0, f(0), f(1), f(2),
                                  f(3), f(4), f(5), stream_of(records)), "be_fold_records_f32");
    std::vector<Tensor> r;
    for (int i = 0; i < 6; ++i) if (t[i].defined()) r.push_back(t[i]);
    return r;
}

// LocalLoss forward + analytic backward in one launch -> (partial [B,3], grad_est [B,10] or an empty tensor)
std::tuple<Tensor, Tensor> local_loss(const Tensor& opts, const Tensor& est, const Tensor& img_fit, const Tensor& gt, const Tensor& bdist,
                                      const Tensor& deri, double beta_b, double beta_s, bool want_grad) {
    TORCH_CHECK(est.dim() == 2 && est.size(1) == 10, "local_loss: est [B,10]");
    fp(est, "est");
    const int64_t b = est.size(0);
    TORCH_CHECK(img_fit.numel() == b * BE_NPIX * 3 && gt.numel() == b * BE_NPIX * 3 && bdist.numel() == b * BE_NPIX && deri.numel() == b * 361 * 3,
                "local_loss: img / gt [B,21,21,3], bdist [B,21,21], deri [B,19,19,3]");
    Tensor partial = at::empty({b, 3}, est.options());
    Tensor grad = want_grad ? at::empty({b, 10}, est.options()) : at::empty({0}, est.options());
    check(be_local_loss_f32(host_struct<be_render_opts>(opts, "local_loss(opts)"), est.data_ptr<float>(), fp(img_fit, "img_fit"), fp(gt, "gt"),
                            fp(bdist, "bdist"), fp(deri, "deri"), (float)beta_b, (float)beta_s, partial.data_ptr<float>(),
                            want_grad ? grad.data_ptr<float>() : nullptr, nullptr, nullptr, b, stream_of(est)), "be_local_loss_f32");
    return {partial, grad};
}
Tensor local_loss_finish(const Tensor& partial, double beta_b, double beta_s) {
    TORCH_CHECK(partial.dim() == 2 && partial.size(1) == 3, "local_loss_finish: partial [B,3]");
    Tensor out = at::empty({1}, partial.options());
    check(be_local_loss_finish_f32(fp(partial, "partial"), (int)partial.size(0), (float)beta_b, (float)beta_s, out.data_ptr<float>(),
                                   stream_of(partial)), "be_local_loss_finish_f32");
    return out;
}

// GlobalLoss terms + analytic gradient in one launch -> (partial [B*P,8], grad [B*P,12], grad_depth [B*P,4])
std::tuple<Tensor, Tensor, Tensor> global_loss(const Tensor& opts, const Tensor& consts, const Tensor& est, const Tensor& img_fit, const Tensor& img_gt,
                                               const Tensor& G, const Tensor& Gd, const Tensor& Gb, const Tensor& bdist, const Tensor& deri,
                                               const Tensor& bdepth, at::ArrayRef<double> gamma6, int64_t hp, int64_t wp, int64_t stride) {
    TORCH_CHECK(est.dim() == 3 && est.size(2) == 12 && est.size(1) == hp * wp, "global_loss: est [B,P,12] with P = hp * wp");
    TORCH_CHECK(img_gt.dim() == 5 && gamma6.size() == 6, "global_loss: img_gt [B,2,H,W,3], six gammas");
    fp(est, "est");
    const int64_t B = est.size(0), P = est.size(1), H = img_gt.size(2), W = img_gt.size(3);
    float g6[6];
    for (int i = 0; i < 6; ++i) g6[i] = (float)gamma6[i];
    Tensor partial = at::empty({B * P, 8}, est.options()), grad = at::empty({B * P, 12}, est.options()), gdep = at::empty({B * P, 4}, est.options());
    check(be_global_loss_f32(host_struct<be_render_opts>(opts, "global_loss(opts)"), host_struct<be_depth_consts>(consts, "global_loss(consts)"),
                             est.data_ptr<float>(), fp(img_fit, "img_fit"), fp(img_gt, "img_gt"), fp(G, "G"), fp(Gd, "Gderi"), fp(Gb, "Gbndry"),
                             fp(bdist, "bdist"), fp(deri, "deri"), fp(bdepth, "bdepth"), g6, partial.data_ptr<float>(), grad.data_ptr<float>(),
                             gdep.data_ptr<float>(), (int)B, (int)hp, (int)wp, (int)H, (int)W, (int)stride, stream_of(est)), "be_global_loss_f32");
    return {partial, grad, gdep};
}

// multi-head self-attention (head dim 16), inference / training forward / backward; workspaces are the caller's (reused across layers)
std::tuple<Tensor, Tensor> attention(const Tensor& qkv, int64_t B, int64_t L, int64_t l_valid, int64_t H, const optional<Tensor>& workspace_) {
    fp(qkv, "qkv");
    const size_t need = be_attention_workspace_floats((int)B, (int)L, (int)H);
    Tensor ws = workspace_.has_value() && workspace_->defined() && (size_t)workspace_->numel() >= need ? *workspace_
                                                                                                        : at::empty({(int64_t)need}, qkv.options());
    Tensor out = at::empty({B * L, H * 16}, qkv.options());
    check(be_attention_f32(qkv.data_ptr<float>(), out.data_ptr<float>(), fpm(ws, "workspace"), (int)B, (int)L, (int)l_valid, (int)H, stream_of(qkv)),
          "be_attention_f32");
    return {out, ws};
}
std::tuple<Tensor, Tensor> attention_train_fwd(const Tensor& qkv, Tensor workspace, int64_t B, int64_t L, int64_t l_valid, int64_t H, double p,
                                               int64_t seed) {
    fp(qkv, "qkv");
    TORCH_CHECK((size_t)workspace.numel() >= be_attention_train_workspace_floats((int)B, (int)L, (int)H), "attention_train_fwd: workspace too small");
    Tensor out = at::empty({B * L, H * 16}, qkv.options()), lse = at::empty({B * H, L}, qkv.options());
    check(be_attention_train_fwd_f32(qkv.data_ptr<float>(), out.data_ptr<float>(), lse.data_ptr<float>(), fpm(workspace, "workspace"), (int)B, (int)L,
                                     (int)l_valid, (int)H, (float)p, (uint32_t)seed, stream_of(qkv)), "be_attention_train_fwd_f32");
    return {out, lse};
}
Tensor attention_bwd(const Tensor& qkv, const Tensor& out, const Tensor& lse, const Tensor& dout, Tensor workspace, Tensor scratch,
                     bool operands_ready, int64_t B, int64_t L, int64_t l_valid, int64_t H, double p, int64_t seed) {
    fp(qkv, "qkv");
    TORCH_CHECK((size_t)workspace.numel() >= be_attention_train_workspace_floats((int)B, (int)L, (int)H) &&
                (size_t)scratch.numel() >= be_attention_bwd_scratch_floats((int)B, (int)L, (int)H), "attention_bwd: workspace / scratch too small");
    Tensor dqkv = at::empty_like(qkv);
    check(be_attention_bwd_f32(qkv.data_ptr<float>(), fp(out, "out"), fp(lse, "lse"), fp(dout, "dout"), dqkv.data_ptr<float>(),
                               fpm(workspace, "workspace"), fpm(scratch, "scratch"), operands_ready ? 1 : 0, (int)B, (int)L, (int)l_valid, (int)H,
                               (float)p, (uint32_t)seed, stream_of(qkv)), "be_attention_bwd_f32");
    return dqkv;
}

}  // namespace

TORCH_LIBRARY(be, m) {
    m.def("local_stage_pack(Tensor[] tensors, float eps) -> Tensor");
    m.def("local_stage_forward(Tensor packed, Tensor x, Tensor(a!)? out, Tensor(b!)? workspace, bool winograd, int chunk) -> (Tensor(a!), Tensor(b!))");
    m.def("render_colors(Tensor opts, Tensor params10, Tensor patches, Tensor(a!)? colors) -> Tensor(a!)");
    m.def("local_depth(Tensor consts, Tensor params10, Tensor(a!)? out) -> Tensor(a!)");
    m.def("train_unit_fwd(Tensor x, Tensor pw, Tensor pb, Tensor gamma, Tensor beta, Tensor? res, Tensor(a!) run_mean, Tensor(b!) run_var, "
          "int cout, int ksize, bool act, float eps, float momentum, Tensor(c!) scratch) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("train_unit_bwd(Tensor x, Tensor dout, Tensor? s_in, Tensor y, Tensor mean, Tensor invstd, Tensor gamma, Tensor? dg_pw, Tensor? dg_pb, "
          "Tensor? dx_add, int ksize, int chw_hw, Tensor(a!) dgamma, Tensor(b!) dbeta, Tensor(c!) dw, Tensor(d!) db, Tensor(e!) scratch) -> (Tensor, Tensor)");
    m.def("train_unit_pair_fwd(Tensor x, Tensor pw_a, Tensor pb_a, Tensor gamma_a, Tensor beta_a, Tensor(a!) rm_a, Tensor(b!) rv_a, int cout_a, "
          "int ks_a, bool act_a, Tensor pw_b, Tensor pb_b, Tensor gamma_b, Tensor beta_b, Tensor(c!) rm_b, Tensor(d!) rv_b, int cout_b, int ks_b, "
          "bool act_b, float eps, float momentum, Tensor(e!) scratch) -> Tensor[]");
    m.def("train_unit_pair_bwd(Tensor x, Tensor dout_a, Tensor? s_in_a, Tensor y_a, Tensor mean_a, Tensor invstd_a, Tensor gamma_a, Tensor dg_pw_a, "
          "Tensor dg_pb_a, int ks_a, Tensor(a!) dgamma_a, Tensor(b!) dbeta_a, Tensor(c!) dw_a, Tensor(d!) db_a, Tensor dout_b, Tensor? s_in_b, "
          "Tensor y_b, Tensor mean_b, Tensor invstd_b, Tensor gamma_b, Tensor dg_pw_b, Tensor dg_pb_b, int ks_b, Tensor(e!) dgamma_b, "
          "Tensor(f!) dbeta_b, Tensor(g!) dw_b, Tensor(h!) db_b, Tensor(i!) scratch) -> Tensor[]");
    m.def("maxpool_fwd_idx(Tensor x, int k, int stride, int pad) -> (Tensor, Tensor)");
    m.def("maxpool_bwd_idx(Tensor idx, Tensor dout, int h, int w, int k, int stride, int pad) -> Tensor");
    m.def("clip_adamw(Tensor table, int nentries, Tensor(a!) grad_flat, Tensor(b!) partial, float max_norm, float grad_scale, float lr, float beta1, "
          "float beta2, float eps, float weight_decay, Tensor(c!) step, Tensor(d!) grad_norm, bool write_back) -> ()");
    m.def("params2etas(Tensor p) -> Tensor");
    m.def("params2etas_bwd(Tensor p, Tensor geta) -> Tensor");
    m.def("params2dists(Tensor opts, Tensor params8) -> Tensor");
    m.def("params2dists_bwd(Tensor opts, Tensor params8, Tensor gdists) -> Tensor");
    m.def("dists2indicators(Tensor dists, Tensor etas) -> Tensor");
    m.def("dists2indicators_bwd(Tensor dists, Tensor etas, Tensor gwedges) -> (Tensor, Tensor)");
    m.def("inverse3x3(Tensor a) -> Tensor");
    m.def("inverse3x3_bwd(Tensor inv, Tensor gout) -> Tensor");
    m.def("image_derivative(Tensor img) -> Tensor");
    m.def("image_derivative_bwd(Tensor img, Tensor gout) -> Tensor");
    m.def("normalized_gaussian(Tensor x, float delta_sq) -> Tensor");
    m.def("normalized_gaussian_bwd(Tensor x, Tensor gy, float delta_sq) -> Tensor");
    m.def("etas2depth(Tensor consts, Tensor eta1, Tensor eta2) -> Tensor");
    m.def("etas2depth_bwd(Tensor consts, Tensor eta1, Tensor eta2, Tensor gz) -> (Tensor, Tensor)");
    m.def("depth2sigma(Tensor consts, Tensor depth, float rho_prime) -> Tensor");
    m.def("depth2sigma_bwd(Tensor consts, Tensor depth, float rho_prime, Tensor geta) -> Tensor");
    m.def("fold_patches(Tensor src, int B, int C, int hp, int wp, int H, int W, int stride, int mode) -> Tensor");
    m.def("fold_patches_bwd(Tensor gout, int hp, int wp, int stride, int mode) -> Tensor");
    m.def("wrap_angles_(Tensor(a!) est, int col0, int col1) -> ()");
    m.def("render_full(Tensor opts, Tensor consts, float rho_prime, bool densify_w, Tensor params12, Tensor view, Tensor pixels, int want) -> Tensor[]");
    m.def("fold_records(Tensor opts, Tensor records, int hp, int wp, int H, int W, int stride, bool densify_w, int want) -> Tensor[]");
    m.def("local_loss(Tensor opts, Tensor est, Tensor img_fit, Tensor gt, Tensor bdist, Tensor deri, float beta_b, float beta_s, bool want_grad) -> (Tensor, Tensor)");
    m.def("local_loss_finish(Tensor partial, float beta_b, float beta_s) -> Tensor");
    m.def("global_loss(Tensor opts, Tensor consts, Tensor est, Tensor img_fit, Tensor img_gt, Tensor G, Tensor Gd, Tensor Gb, Tensor bdist, Tensor deri, "
          "Tensor bdepth, float[] gamma6, int hp, int wp, int stride) -> (Tensor, Tensor, Tensor)");
    m.def("attention(Tensor qkv, int B, int L, int l_valid, int H, Tensor(a!)? workspace) -> (Tensor, Tensor(a!))");
    m.def("attention_train_fwd(Tensor qkv, Tensor(a!) workspace, int B, int L, int l_valid, int H, float p, int seed) -> (Tensor, Tensor)");
    m.def("attention_bwd(Tensor qkv, Tensor out, Tensor lse, Tensor dout, Tensor(a!) workspace, Tensor(b!) scratch, bool operands_ready, int B, int L, "
          "int l_valid, int H, float p, int seed) -> Tensor");
}

// on ROCm builds of PyTorch the GPU dispatch key is named CUDA (HIP masquerades as it); ops that take only host structs + GPU tensors
// dispatch on the GPU tensors
TORCH_LIBRARY_IMPL(be, CUDA, m) {
    m.impl("local_stage_pack", local_stage_pack);
    m.impl("local_stage_forward", local_stage_forward);
    m.impl("train_unit_fwd", train_unit_fwd);
    m.impl("train_unit_bwd", train_unit_bwd);
    m.impl("train_unit_pair_fwd", train_unit_pair_fwd);
    m.impl("train_unit_pair_bwd", train_unit_pair_bwd);
    m.impl("maxpool_fwd_idx", maxpool_fwd_idx);
    m.impl("maxpool_bwd_idx", maxpool_bwd_idx);
    m.impl("clip_adamw", clip_adamw);
}
// registered for every backend: these check their GPU arguments themselves, so that a CPU tensor raises the library's own
// "no CPU fallback" error instead of the dispatcher's missing-kernel message (some also take a CPU byte tensor - the host struct -
// next to their GPU tensors)
TORCH_LIBRARY_IMPL(be, CompositeExplicitAutograd, m) {
    m.impl("render_colors", render_colors);
    m.impl("local_depth", local_depth);
    m.impl("params2dists", params2dists);
    m.impl("params2dists_bwd", params2dists_bwd);
    m.impl("etas2depth", etas2depth);
    m.impl("etas2depth_bwd", etas2depth_bwd);
    m.impl("depth2sigma", depth2sigma);
    m.impl("depth2sigma_bwd", depth2sigma_bwd);
    m.impl("params2etas", params2etas);
    m.impl("params2etas_bwd", params2etas_bwd);
    m.impl("dists2indicators", dists2indicators);
    m.impl("dists2indicators_bwd", dists2indicators_bwd);
    m.impl("inverse3x3", inverse3x3);
    m.impl("inverse3x3_bwd", inverse3x3_bwd);
    m.impl("image_derivative", image_derivative);
    m.impl("image_derivative_bwd", image_derivative_bwd);
    m.impl("normalized_gaussian", normalized_gaussian);
    m.impl("normalized_gaussian_bwd", normalized_gaussian_bwd);
    m.impl("fold_patches", fold_patches);
    m.impl("fold_patches_bwd", fold_patches_bwd);
    m.impl("wrap_angles_", wrap_angles_);
    m.impl("render_full", render_full);
    m.impl("fold_records", fold_records);
    m.impl("local_loss", local_loss);
    m.impl("local_loss_finish", local_loss_finish);
    m.impl("global_loss", global_loss);
    m.impl("attention", attention);
    m.impl("attention_train_fwd", attention_train_fwd);
    m.impl("attention_bwd", attention_bwd);
}
