// ops.h, third batch (round 4): multi-axis and whole-array reductions, variance, scans, top-k / partition, triangles and identities,
// linspace, outer / inner, isclose / allclose / array_equal, pad / repeat / tile / diag(onal), nan_to_num, broadcast_arrays -- composed
// from the ops of mlxc.hip / mlxc_glue.hpp (same translation unit) plus two kernels (product, scan).  Not on the four callers' path;
// `mlx-rs` names them (mlx-rs/mlx-sys/src/mlx-c/mlx/c/ops.h; signatures cited per function in include/omx_mlx_c.h).
#pragma once

namespace {

struct Tmp {   // a temporary handle, freed at scope exit
    mlx_array a = mlx_array_new();
    ~Tmp() { mlx_array_free(a); }
    mlx_array* operator&() { return &a; }
    operator mlx_array() const { return a; }
};

// product over the middle axis of a contiguous [outer, n, inner]
__global__ void prod_axis_kernel(char* out, int odt, const char* in, int idt, size_t outer, int n, size_t inner) {
    const size_t total = outer * inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / inner, j = i % inner;
        if (is_int_dt(idt)) {
            long long acc = 1;
            for (int k = 0; k < n; ++k) acc *= ld_i(in, idt, (o * n + k) * inner + j);
            st_i(out, odt, i, acc);
        } else {
            float acc = 1.f;
            for (int k = 0; k < n; ++k) acc *= ld_f(in, idt, (o * n + k) * inner + j);
            st_f(out, odt, i, acc);
        }
    }
}
// running sum / product / max / min along the middle axis (mode 0..3), optionally from the end, optionally exclusive
__global__ void scan_axis_kernel(char* out, const char* in, int dt, size_t outer, int n, size_t inner, int mode, bool reverse, bool inclusive) {
    const size_t total = outer * inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / inner, j = i % inner;
        if (is_int_dt(dt)) {
            long long acc = mode == 0 ? 0 : mode == 1 ? 1 : mode == 2 ? LLONG_MIN : LLONG_MAX;
            for (int t = 0; t < n; ++t) {
                const int k = reverse ? n - 1 - t : t;
                const size_t at = (o * n + k) * inner + j;
                const long long v = ld_i(in, dt, at);
                const long long nxt = mode == 0 ? acc + v : mode == 1 ? acc * v : mode == 2 ? (v > acc ? v : acc) : (v < acc ? v : acc);
                st_i(out, dt, at, inclusive ? nxt : acc);
                acc = nxt;
            }
        } else {
            float acc = mode == 0 ? 0.f : mode == 1 ? 1.f : mode == 2 ? -INFINITY : INFINITY;
            for (int t = 0; t < n; ++t) {
                const int k = reverse ? n - 1 - t : t;
                const size_t at = (o * n + k) * inner + j;
                const float v = ld_f(in, dt, at);
                const float nxt = mode == 0 ? acc + v : mode == 1 ? acc * v : (v != v || acc != acc) ? NAN : mode == 2 ? fmaxf(acc, v) : fminf(acc, v);
                st_f(out, dt, at, inclusive ? nxt : acc);
                acc = nxt;
            }
        }
    }
}

typedef int (*axis_reduce_fn)(mlx_array*, const mlx_array, int, bool, const mlx_stream);

// the reduced axes moved to the end and flattened into one: [kept..., prod(reduced)]; keep_shape = the input's shape with 1 at the reduced axes
int fold_axes(const mlx_array a, const int* axes, size_t n_axes, const char* name, mlx_array* folded, std::vector<int>* keep_shape,
              std::vector<int>* perm_out, std::vector<int>* moved_shape, const mlx_stream s) {
    const Arr& src = *A(a);
    const int nd = (int)src.shape.size();
    std::vector<bool> red((size_t)nd, false);
    for (size_t i = 0; i < n_axes; ++i) {
        int ax;
        if (norm_axis(axes[i], nd, name, &ax)) return 1;
        OMX_REQUIRE(!red[ax], "%s: axis %d given twice", name, axes[i]);
        red[ax] = true;
    }
    std::vector<int> perm, kept, shape_after;
    int prod = 1;
    for (int i = 0; i < nd; ++i)
        if (!red[i]) { perm.push_back(i); kept.push_back(src.shape[i]); }
    for (int i = 0; i < nd; ++i)
        if (red[i]) { perm.push_back(i); prod *= src.shape[i]; }
    *keep_shape = src.shape;
    for (int i = 0; i < nd; ++i)
        if (red[i]) (*keep_shape)[i] = 1;
    if (perm_out) *perm_out = perm;
    if (moved_shape) {
        moved_shape->clear();
        for (int p : perm) moved_shape->push_back(src.shape[p]);
    }
    Tmp t;
    if (mlx_transpose_axes(&t, a, perm.data(), perm.size(), s)) return 1;
    shape_after = kept;
    shape_after.push_back(prod);
    return mlx_reshape(folded, t, shape_after.data(), shape_after.size(), s);
}

int reduce_axes(mlx_array* res, const mlx_array a, const int* axes, size_t n_axes, bool keepdims, const mlx_stream s, axis_reduce_fn fn, const char* name) {
    REQ_ARR(a, name);
    if (n_axes == 0) return mlx_array_set(res, a);
    Tmp folded, red;
    std::vector<int> keep_shape;
    if (fold_axes(a, axes, n_axes, name, &folded, &keep_shape, nullptr, nullptr, s)) return 1;
    if (fn(&red, folded, -1, false, s)) return 1;
    if (keepdims) return mlx_reshape(res, red, keep_shape.data(), keep_shape.size(), s);
    return mlx_array_set(res, red);
}

std::vector<int> all_axes_of(const mlx_array a) {
    std::vector<int> ax(A(a)->shape.size());
    for (size_t i = 0; i < ax.size(); ++i) ax[i] = (int)i;
    return ax;
}

mlx_array f32_scalar(float v) { return mlx_array_new_float(v); }


// Conv2d, channels-last like MLX: input [B, H, W, Cin], weight [Cout, kH, kW, Cin / groups], output [B, Ho, Wo, Cout]; one thread per
// output element, f32 accumulation in (kh, kw, ci) order -- the general form (the bf16 1x1 / 3x3 shapes of the VAE take the GEMM routes)
struct Conv2dGeom { int B, H, W, Cin, Ho, Wo, Cout, kH, kW, s0, s1, p0, p1, d0, d1, groups; };
__global__ void conv2d_kernel(char* out, const char* x, const char* w, int dt, const Conv2dGeom g) {
    const size_t total = (size_t)g.B * g.Ho * g.Wo * g.Cout;
    const int cig = g.Cin / g.groups, cog = g.Cout / g.groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % g.Cout), ox = (int)((i / g.Cout) % g.Wo), oy = (int)((i / ((size_t)g.Cout * g.Wo)) % g.Ho);
        const int b = (int)(i / ((size_t)g.Cout * g.Wo * g.Ho)), grp = co / cog;
        float acc = 0.f;
        for (int kh = 0; kh < g.kH; ++kh) {
            const int y = oy * g.s0 - g.p0 + kh * g.d0;
            if (y < 0 || y >= g.H) continue;
            for (int kw = 0; kw < g.kW; ++kw) {
                const int xx = ox * g.s1 - g.p1 + kw * g.d1;
                if (xx < 0 || xx >= g.W) continue;
                const size_t xi = (((size_t)b * g.H + y) * g.W + xx) * g.Cin + grp * cig, wi = (((size_t)co * g.kH + kh) * g.kW + kw) * cig;
                for (int ci = 0; ci < cig; ++ci) acc = fmaf(ld_f(x, dt, xi + ci), ld_f(w, dt, wi + ci), acc);
            }
        }
        st_f(out, dt, i, acc);
    }
}
}  // namespace

extern "C" {

int mlx_prod_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) {
    REQ_ARR(a, "mlx_prod_axis");
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), "mlx_prod_axis", &ax)) return 1;
    Contig c;
    if (c.init(s)) return 1;
    size_t outer, inner; int n;
    around_axis(s.shape, ax, &outer, &n, &inner);
    std::vector<int> shape = s.shape;
    if (keepdims) shape[ax] = 1; else shape.erase(shape.begin() + ax);
    const mlx_dtype odt = s.dt == MLX_BOOL ? MLX_INT32 : s.dt;
    NEW_OR_FAIL(r, shape, odt);
    if (r->size()) {
        prod_axis_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), odt, c.a->ptr(), s.dt, outer, n, inner);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
#define OMX_AXES_FORM(NAME, AXIS_FN)                                                                                                \
    int NAME(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s) {              \
        return reduce_axes(res, a, axes, axes_num, keepdims, s, AXIS_FN, #NAME);                                                   \
    }
OMX_AXES_FORM(mlx_sum_axes, mlx_sum_axis) OMX_AXES_FORM(mlx_mean_axes, mlx_mean_axis) OMX_AXES_FORM(mlx_max_axes, mlx_max_axis)
OMX_AXES_FORM(mlx_min_axes, mlx_min_axis) OMX_AXES_FORM(mlx_all_axes, mlx_all_axis) OMX_AXES_FORM(mlx_any_axes, mlx_any_axis)
OMX_AXES_FORM(mlx_logsumexp_axes, mlx_logsumexp_axis) OMX_AXES_FORM(mlx_prod_axes, mlx_prod_axis)
#undef OMX_AXES_FORM
int mlx_prod(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) {
    REQ_ARR(a, "mlx_prod");
    const std::vector<int> ax = all_axes_of(a);
    if (ax.empty()) return mlx_array_set(res, a);
    return mlx_prod_axes(res, a, ax.data(), ax.size(), keepdims, s);
}

// var = sum((a - mean)^2) / (n - ddof) over the axes, in the input's float dtype (integers: float32)
int mlx_var_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, int ddof, const mlx_stream s) {
    REQ_ARR(a, "mlx_var_axes");
    Tmp af, mu, dev, sq, ssum, denom;
    const mlx_dtype fdt = is_float(A(a)->dt) ? A(a)->dt : MLX_FLOAT32;
    if (mlx_astype(&af, a, fdt, s)) return 1;
    if (mlx_mean_axes(&mu, af, axes, axes_num, true, s) || mlx_subtract(&dev, af, mu, s) || mlx_multiply(&sq, dev, dev, s)) return 1;
    if (mlx_sum_axes(&ssum, sq, axes, axes_num, keepdims, s)) return 1;
    const size_t n_in = A(a)->size(), n_out = A(ssum.a)->size();
    const double n = n_out ? (double)n_in / (double)n_out : 0.0;
    Tmp d32;
    d32.a = f32_scalar((float)std::max(n - (double)ddof, 0.0));
    if (mlx_astype(&denom, d32, fdt, s)) return 1;
    return mlx_divide(res, ssum, denom, s);
}
int mlx_var_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, int ddof, const mlx_stream s) { return mlx_var_axes(res, a, &axis, 1, keepdims, ddof, s); }
int mlx_var(mlx_array* res, const mlx_array a, bool keepdims, int ddof, const mlx_stream s) {
    REQ_ARR(a, "mlx_var");
    const std::vector<int> ax = all_axes_of(a);
    return mlx_var_axes(res, a, ax.data(), ax.size(), keepdims, ddof, s);
}
int mlx_std_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, int ddof, const mlx_stream s) {
    Tmp v;
    if (mlx_var_axes(&v, a, axes, axes_num, keepdims, ddof, s)) return 1;
    return mlx_sqrt(res, v, s);
}
int mlx_std_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, int ddof, const mlx_stream s) { return mlx_std_axes(res, a, &axis, 1, keepdims, ddof, s); }
int mlx_std(mlx_array* res, const mlx_array a, bool keepdims, int ddof, const mlx_stream s) {
    REQ_ARR(a, "mlx_std");
    const std::vector<int> ax = all_axes_of(a);
    return mlx_std_axes(res, a, ax.data(), ax.size(), keepdims, ddof, s);
}

// softmax over several axes: over their flattened product, then back to the input's layout
int mlx_softmax_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool precise, const mlx_stream s) {
    REQ_ARR(a, "mlx_softmax_axes");
    if (axes_num == 0) return mlx_array_set(res, a);
    Tmp folded, sm, unfolded;
    std::vector<int> keep_shape, perm, moved_shape;
    if (fold_axes(a, axes, axes_num, "mlx_softmax_axes", &folded, &keep_shape, &perm, &moved_shape, s)) return 1;
    if (mlx_softmax_axis(&sm, folded, -1, precise, s)) return 1;
    if (mlx_reshape(&unfolded, sm, moved_shape.data(), moved_shape.size(), s)) return 1;
    std::vector<int> inv(perm.size());
    for (size_t i = 0; i < perm.size(); ++i) inv[perm[i]] = (int)i;
    return mlx_transpose_axes(res, unfolded, inv.data(), inv.size(), s);
}
int mlx_softmax(mlx_array* res, const mlx_array a, bool precise, const mlx_stream s) {
    REQ_ARR(a, "mlx_softmax");
    const std::vector<int> ax = all_axes_of(a);
    return mlx_softmax_axes(res, a, ax.data(), ax.size(), precise, s);
}

// argmin along an axis: position 0 of the stable ascending argsort (ties: the lowest index, like MLX); the whole-array forms flatten
int mlx_argmin_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s) {
    REQ_ARR(a, "mlx_argmin_axis");
    const int nd = (int)A(a)->shape.size();
    int ax;
    if (norm_axis(axis, nd, "mlx_argmin_axis", &ax)) return 1;
    Tmp order, first;
    if (mlx_argsort_axis(&order, a, ax, s)) return 1;
    std::vector<int> start((size_t)nd, 0), stop = A(a)->shape, strides((size_t)nd, 1);
    stop[ax] = 1;
    if (mlx_slice(&first, order, start.data(), nd, stop.data(), nd, strides.data(), nd, s)) return 1;
    if (keepdims) return mlx_array_set(res, first);
    return mlx_squeeze_axis(res, first, ax, s);
}
static int arg_all(mlx_array* res, const mlx_array a, bool keepdims, bool is_max, const mlx_stream s) {
    const int nd = (int)A(a)->shape.size();
    Tmp flat, idx;
    int rc = nd == 1 ? mlx_array_set(&flat, a) : nd == 0 ? mlx_reshape(&flat, a, std::vector<int>{1}.data(), 1, s) : mlx_flatten(&flat, a, 0, -1, s);
    if (!rc) rc = is_max ? mlx_argmax_axis(&idx, flat, 0, false, s) : mlx_argmin_axis(&idx, flat, 0, false, s);
    if (rc) return 1;
    if (keepdims && nd > 0) {
        std::vector<int> ones((size_t)nd, 1);
        return mlx_reshape(res, idx, ones.data(), ones.size(), s);
    }
    return mlx_array_set(res, idx);
}
int mlx_argmax(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { REQ_ARR(a, "mlx_argmax"); return arg_all(res, a, keepdims, true, s); }
int mlx_argmin(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { REQ_ARR(a, "mlx_argmin"); return arg_all(res, a, keepdims, false, s); }

static int scan_axis(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, int mode, const char* name) {
    REQ_ARR(a, name);
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), name, &ax)) return 1;
    Contig c;
    if (c.init(s)) return 1;
    size_t outer, inner; int n;
    around_axis(s.shape, ax, &outer, &n, &inner);
    const mlx_dtype odt = s.dt == MLX_BOOL ? MLX_INT32 : s.dt;
    OMX_REQUIRE(odt == s.dt, "%s: boolean input is not supported (cast it first)", name);
    NEW_OR_FAIL(r, s.shape, odt);
    if (r->size()) {
        scan_axis_kernel<<<grid_for(outer * inner), 256, 0, g_stream>>>(r->ptr(), c.a->ptr(), s.dt, outer, n, inner, mode, reverse, inclusive);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_cumsum(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream) { return scan_axis(res, a, axis, reverse, inclusive, 0, "mlx_cumsum"); }
int mlx_cumprod(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream) { return scan_axis(res, a, axis, reverse, inclusive, 1, "mlx_cumprod"); }
int mlx_cummax(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream) { return scan_axis(res, a, axis, reverse, inclusive, 2, "mlx_cummax"); }
int mlx_cummin(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream) { return scan_axis(res, a, axis, reverse, inclusive, 3, "mlx_cummin"); }

// partition / top-k: MLX promises the kth element in its sorted place, nothing about the order inside the two sides -- a full sort
// is one valid answer (as for mlx_argpartition_axis); top-k = the last k of it
int mlx_partition_axis(mlx_array* res, const mlx_array a, int kth, int axis, const mlx_stream s) {
    REQ_ARR(a, "mlx_partition_axis");
    int ax;
    if (norm_axis(axis, (int)A(a)->shape.size(), "mlx_partition_axis", &ax)) return 1;
    const int n = A(a)->shape[ax], k = kth < 0 ? kth + n : kth;
    OMX_REQUIRE(k >= 0 && k < n, "mlx_partition_axis: kth %d out of range for %d elements", kth, n);
    return mlx_sort_axis(res, a, ax, s);
}
int mlx_partition(mlx_array* res, const mlx_array a, int kth, const mlx_stream s) {
    REQ_ARR(a, "mlx_partition");
    Tmp flat;
    if (A(a)->shape.size() <= 1 ? mlx_array_set(&flat, a) : mlx_flatten(&flat, a, 0, -1, s)) return 1;
    return mlx_partition_axis(res, flat, kth, 0, s);
}
int mlx_argpartition(mlx_array* res, const mlx_array a, int kth, const mlx_stream s) {
    REQ_ARR(a, "mlx_argpartition");
    Tmp flat;
    if (A(a)->shape.size() <= 1 ? mlx_array_set(&flat, a) : mlx_flatten(&flat, a, 0, -1, s)) return 1;
    return mlx_argpartition_axis(res, flat, kth, 0, s);
}
int mlx_topk_axis(mlx_array* res, const mlx_array a, int k, int axis, const mlx_stream s) {
    REQ_ARR(a, "mlx_topk_axis");
    const int nd = (int)A(a)->shape.size();
    int ax;
    if (norm_axis(axis, nd, "mlx_topk_axis", &ax)) return 1;
    const int n = A(a)->shape[ax];
    OMX_REQUIRE(k >= 0 && k <= n, "mlx_topk_axis: k %d out of range for %d elements", k, n);
    Tmp sorted;
    if (mlx_sort_axis(&sorted, a, ax, s)) return 1;
    std::vector<int> start((size_t)nd, 0), stop = A(a)->shape, strides((size_t)nd, 1);
    start[ax] = n - k;
    return mlx_slice(res, sorted, start.data(), nd, stop.data(), nd, strides.data(), nd, s);
}
int mlx_topk(mlx_array* res, const mlx_array a, int k, const mlx_stream s) {
    REQ_ARR(a, "mlx_topk");
    return mlx_topk_axis(res, a, k, -1, s);
}

// tri(n, m, k)[i][j] = j <= i + k; eye(n, m, k)[i][j] = j == i + k
static int index_mask(mlx_array* res, int n, int m, int k, bool equal, const mlx_stream s) {
    OMX_REQUIRE(n >= 0 && m >= 0, "tri / eye: negative size");
    Tmp rows, cols, rows2, cols2, kk, shifted;
    if (mlx_arange(&rows, 0, n, 1, MLX_INT32, s) || mlx_arange(&cols, 0, m, 1, MLX_INT32, s)) return 1;
    const int rs[2] = {n, 1}, cs[2] = {1, m};
    if (mlx_reshape(&rows2, rows, rs, 2, s) || mlx_reshape(&cols2, cols, cs, 2, s)) return 1;
    kk.a = mlx_array_new_int(k);
    if (mlx_add(&shifted, rows2, kk, s)) return 1;
    return equal ? mlx_equal(res, cols2, shifted, s) : mlx_less_equal(res, cols2, shifted, s);
}
int mlx_tri(mlx_array* res, int n, int m, int k, mlx_dtype type, const mlx_stream s) {
    Tmp mask;
    if (index_mask(&mask, n, m, k, false, s)) return 1;
    return mlx_astype(res, mask, type, s);
}
int mlx_eye(mlx_array* res, int n, int m, int k, mlx_dtype dtype, const mlx_stream s) {
    Tmp mask;
    if (index_mask(&mask, n, m, k, true, s)) return 1;
    return mlx_astype(res, mask, dtype, s);
}
int mlx_identity(mlx_array* res, int n, mlx_dtype dtype, const mlx_stream s) { return mlx_eye(res, n, n, 0, dtype, s); }
static int tri_select(mlx_array* res, const mlx_array x, int k, bool lower, const char* name, const mlx_stream s) {
    REQ_ARR(x, name);
    const Arr& a = *A(x);
    OMX_REQUIRE(a.shape.size() >= 2, "%s: at least two dimensions expected", name);
    const int n = a.shape[a.shape.size() - 2], m = a.shape.back();
    Tmp mask, zero, z;
    if (index_mask(&mask, n, m, lower ? k : k - 1, false, s)) return 1;     // upper: keep where NOT (j <= i + k - 1)
    zero.a = f32_scalar(0.f);
    if (mlx_astype(&z, zero, a.dt, s)) return 1;
    return lower ? mlx_where(res, mask, x, z, s) : mlx_where(res, mask, z, x, s);
}
int mlx_tril(mlx_array* res, const mlx_array x, int k, const mlx_stream s) { return tri_select(res, x, k, true, "mlx_tril", s); }
int mlx_triu(mlx_array* res, const mlx_array x, int k, const mlx_stream s) { return tri_select(res, x, k, false, "mlx_triu", s); }

// num evenly spaced values from start to stop inclusive: start + i * (stop - start) / (num - 1) in float32, then the dtype
int mlx_linspace(mlx_array* res, double start, double stop, int num, mlx_dtype dtype, const mlx_stream s) {
    OMX_REQUIRE(num >= 0, "mlx_linspace: num must be >= 0 (got %d)", num);
    Tmp idx, step, scaled, st0, out32;
    if (mlx_arange(&idx, 0, num, 1, MLX_FLOAT32, s)) return 1;
    step.a = f32_scalar(num > 1 ? (float)((stop - start) / (double)(num - 1)) : 0.f);
    st0.a = f32_scalar((float)start);
    if (mlx_multiply(&scaled, idx, step, s) || mlx_add(&out32, scaled, st0, s)) return 1;
    return mlx_astype(res, out32, dtype, s);
}

int mlx_outer(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s) {
    REQ_ARR(a, "mlx_outer"); REQ_ARR(b, "mlx_outer");
    Tmp col, row;
    const int cs[2] = {(int)A(a)->size(), 1}, rs[2] = {1, (int)A(b)->size()};
    if (mlx_reshape(&col, a, cs, 2, s) || mlx_reshape(&row, b, rs, 2, s)) return 1;
    return mlx_multiply(res, col, row, s);
}
// inner product over the last axes: (..., K) x (..., K) -> sum over K of the outer combination of the leading axes (numpy.inner)
int mlx_inner(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s) {
    REQ_ARR(a, "mlx_inner"); REQ_ARR(b, "mlx_inner");
    const Arr &x = *A(a), &y = *A(b);
    if (x.shape.empty() || y.shape.empty()) return mlx_multiply(res, a, b, s);
    OMX_REQUIRE(x.shape.back() == y.shape.back(), "mlx_inner: last dimensions differ (%d vs %d)", x.shape.back(), y.shape.back());
    const int K = x.shape.back();
    Tmp x2, y2, prod, summed;
    const int xs[3] = {(int)(x.size() / (size_t)std::max(K, 1)), 1, K}, ys[3] = {1, (int)(y.size() / (size_t)std::max(K, 1)), K};
    if (mlx_reshape(&x2, a, xs, 3, s) || mlx_reshape(&y2, b, ys, 3, s) || mlx_multiply(&prod, x2, y2, s) || mlx_sum_axis(&summed, prod, 2, false, s)) return 1;
    std::vector<int> shape(x.shape.begin(), x.shape.end() - 1);
    shape.insert(shape.end(), y.shape.begin(), y.shape.end() - 1);
    return mlx_reshape(res, summed, shape.data(), shape.size(), s);
}

static int atleast(mlx_array* res, const mlx_array a, int nd_min, const char* name, const mlx_stream s) {
    REQ_ARR(a, name);
    const std::vector<int>& sh = A(a)->shape;
    if ((int)sh.size() >= nd_min) return mlx_array_set(res, a);
    std::vector<int> shape;
    if (nd_min == 1) shape = {1};
    else if (nd_min == 2) shape = sh.empty() ? std::vector<int>{1, 1} : std::vector<int>{1, sh[0]};
    else shape = sh.empty() ? std::vector<int>{1, 1, 1} : sh.size() == 1 ? std::vector<int>{1, sh[0], 1} : std::vector<int>{sh[0], sh[1], 1};   // numpy.atleast_3d
    return mlx_reshape(res, a, shape.data(), shape.size(), s);
}
int mlx_atleast_1d(mlx_array* res, const mlx_array a, const mlx_stream s) { return atleast(res, a, 1, "mlx_atleast_1d", s); }
int mlx_atleast_2d(mlx_array* res, const mlx_array a, const mlx_stream s) { return atleast(res, a, 2, "mlx_atleast_2d", s); }
int mlx_atleast_3d(mlx_array* res, const mlx_array a, const mlx_stream s) { return atleast(res, a, 3, "mlx_atleast_3d", s); }

// |a - b| <= atol + rtol * |b|, or a == b (infinities), or both NaN when equal_nan
int mlx_isclose(mlx_array* res, const mlx_array a, const mlx_array b, double rtol, double atol, bool equal_nan, const mlx_stream s) {
    REQ_ARR(a, "mlx_isclose"); REQ_ARR(b, "mlx_isclose");
    Tmp af, bf, diff, adiff, bb, rt, at, tol0, tol, close0, fa, fb, fab, close, same, both;
    if (mlx_astype(&af, a, MLX_FLOAT32, s) || mlx_astype(&bf, b, MLX_FLOAT32, s)) return 1;
    rt.a = f32_scalar((float)rtol); at.a = f32_scalar((float)atol);
    // (the tolerance test only between finite values: inf - (-inf) <= inf would pass it)
    if (mlx_subtract(&diff, af, bf, s) || mlx_abs(&adiff, diff, s) || mlx_abs(&bb, bf, s) || mlx_multiply(&tol0, bb, rt, s) || mlx_add(&tol, tol0, at, s) ||
        mlx_less_equal(&close0, adiff, tol, s) || mlx_isfinite(&fa, af, s) || mlx_isfinite(&fb, bf, s) || mlx_logical_and(&fab, fa, fb, s) ||
        mlx_logical_and(&close, close0, fab, s) || mlx_equal(&same, af, bf, s) || mlx_logical_or(&both, close, same, s))
        return 1;
    if (!equal_nan) return mlx_array_set(res, both);
    Tmp na, nb, nn;
    if (mlx_isnan(&na, af, s) || mlx_isnan(&nb, bf, s) || mlx_logical_and(&nn, na, nb, s)) return 1;
    return mlx_logical_or(res, both, nn, s);
}
int mlx_allclose(mlx_array* res, const mlx_array a, const mlx_array b, double rtol, double atol, bool equal_nan, const mlx_stream s) {
    Tmp c;
    if (mlx_isclose(&c, a, b, rtol, atol, equal_nan, s)) return 1;
    return mlx_all(res, c, false, s);
}
int mlx_array_equal(mlx_array* res, const mlx_array a, const mlx_array b, bool equal_nan, const mlx_stream s) {
    REQ_ARR(a, "mlx_array_equal"); REQ_ARR(b, "mlx_array_equal");
    if (A(a)->shape != A(b)->shape) { mlx_array f = mlx_array_new_bool(false); const int rc = mlx_array_set(res, f); mlx_array_free(f); return rc; }
    Tmp eq;
    if (mlx_equal(&eq, a, b, s)) return 1;
    if (equal_nan) {
        Tmp na, nb, nn, either;
        if (mlx_isnan(&na, a, s) || mlx_isnan(&nb, b, s) || mlx_logical_and(&nn, na, nb, s) || mlx_logical_or(&either, eq, nn, s)) return 1;
        return mlx_all(res, either, false, s);
    }
    return mlx_all(res, eq, false, s);
}

static int times_scalar(mlx_array* res, const mlx_array a, float f, const char* name, const mlx_stream s) {
    REQ_ARR(a, name);
    const mlx_dtype fdt = is_float(A(a)->dt) ? A(a)->dt : MLX_FLOAT32;
    Tmp af, k32, k;
    k32.a = f32_scalar(f);
    if (mlx_astype(&af, a, fdt, s) || mlx_astype(&k, k32, fdt, s)) return 1;
    return mlx_multiply(res, af, k, s);
}
int mlx_degrees(mlx_array* res, const mlx_array a, const mlx_stream s) { return times_scalar(res, a, (float)(180.0 / 3.14159265358979323846), "mlx_degrees", s); }
int mlx_radians(mlx_array* res, const mlx_array a, const mlx_stream s) { return times_scalar(res, a, (float)(3.14159265358979323846 / 180.0), "mlx_radians", s); }

int mlx_divmod(mlx_vector_array* res, const mlx_array a, const mlx_array b, const mlx_stream s) {
    OMX_REQUIRE(res, "mlx_divmod: null result");
    Tmp q, r;
    if (mlx_floor_divide(&q, a, b, s) || mlx_remainder(&r, a, b, s)) return 1;
    Vec* out = new Vec();
    out->v.push_back(new Arr(*A(q.a)));
    out->v.push_back(new Arr(*A(r.a)));
    for (Arr* p : out->v) p->host.clear();
    delete reinterpret_cast<Vec*>(res->ctx);
    res->ctx = out;
    return 0;
}

int mlx_unflatten(mlx_array* res, const mlx_array a, int axis, const int* shape, size_t shape_num, const mlx_stream s) {
    REQ_ARR(a, "mlx_unflatten");
    int ax;
    if (norm_axis(axis, (int)A(a)->shape.size(), "mlx_unflatten", &ax)) return 1;
    std::vector<int> full(A(a)->shape.begin(), A(a)->shape.begin() + ax);
    long long known = 1;
    int infer = -1;
    for (size_t i = 0; i < shape_num; ++i) {
        if (shape[i] == -1) { OMX_REQUIRE(infer < 0, "mlx_unflatten: more than one -1"); infer = (int)full.size(); full.push_back(1); }
        else { known *= shape[i]; full.push_back(shape[i]); }
    }
    if (infer >= 0) { OMX_REQUIRE(known > 0 && A(a)->shape[ax] % known == 0, "mlx_unflatten: cannot infer the -1 dimension"); full[infer] = (int)(A(a)->shape[ax] / known); known *= full[infer]; }
    OMX_REQUIRE(known == A(a)->shape[ax], "mlx_unflatten: the new shape holds %lld elements, the axis %d", known, A(a)->shape[ax]);
    full.insert(full.end(), A(a)->shape.begin() + ax + 1, A(a)->shape.end());
    return mlx_reshape(res, a, full.data(), full.size(), s);
}

// constant padding: a canvas of the pad value, the input written into its interior
int mlx_pad(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const int* low_pad_size, size_t low_pad_size_num,
            const int* high_pad_size, size_t high_pad_size_num, const mlx_array pad_value, const char* mode, const mlx_stream s) {
    REQ_ARR(a, "mlx_pad"); REQ_ARR(pad_value, "mlx_pad");
    OMX_REQUIRE(!mode || strcmp(mode, "constant") == 0, "mlx_pad: only mode \"constant\" is supported (got \"%s\")", mode);
    OMX_REQUIRE(axes_num == low_pad_size_num && axes_num == high_pad_size_num, "mlx_pad: axes, low and high pad sizes must have one entry per axis");
    const int nd = (int)A(a)->shape.size();
    std::vector<int> shape = A(a)->shape, start((size_t)nd, 0), strides((size_t)nd, 1);
    for (size_t i = 0; i < axes_num; ++i) {
        int ax;
        if (norm_axis(axes[i], nd, "mlx_pad", &ax)) return 1;
        OMX_REQUIRE(low_pad_size[i] >= 0 && high_pad_size[i] >= 0, "mlx_pad: negative pad size");
        shape[ax] += low_pad_size[i] + high_pad_size[i];
        start[ax] = low_pad_size[i];
    }
    std::vector<int> stop((size_t)nd);
    for (int i = 0; i < nd; ++i) stop[i] = start[i] + A(a)->shape[i];
    Tmp canvas;
    if (mlx_full(&canvas, shape.data(), shape.size(), pad_value, A(a)->dt, s)) return 1;
    if (A(a)->size() == 0) return mlx_array_set(res, canvas);
    return mlx_slice_update(res, canvas, a, start.data(), nd, stop.data(), nd, strides.data(), nd, s);
}

int mlx_repeat_axis(mlx_array* res, const mlx_array arr, int repeats, int axis, const mlx_stream s) {
    REQ_ARR(arr, "mlx_repeat_axis");
    OMX_REQUIRE(repeats >= 0, "mlx_repeat_axis: repeats must be >= 0");
    const int nd = (int)A(arr)->shape.size();
    int ax;
    if (norm_axis(axis, nd, "mlx_repeat_axis", &ax)) return 1;
    Tmp ex, bc;
    if (mlx_expand_dims(&ex, arr, ax + 1, s)) return 1;
    std::vector<int> bshape = A(ex.a)->shape;
    bshape[ax + 1] = repeats;
    if (mlx_broadcast_to(&bc, ex, bshape.data(), bshape.size(), s)) return 1;
    std::vector<int> out = A(arr)->shape;
    out[ax] *= repeats;
    return mlx_reshape(res, bc, out.data(), out.size(), s);
}
int mlx_repeat(mlx_array* res, const mlx_array arr, int repeats, const mlx_stream s) {
    REQ_ARR(arr, "mlx_repeat");
    Tmp flat;
    const int one[1] = {(int)A(arr)->size()};
    if (mlx_reshape(&flat, arr, one, 1, s)) return 1;
    return mlx_repeat_axis(res, flat, repeats, 0, s);
}
int mlx_tile(mlx_array* res, const mlx_array arr, const int* reps, size_t reps_num, const mlx_stream s) {
    REQ_ARR(arr, "mlx_tile");
    std::vector<int> shape = A(arr)->shape, r(reps, reps + reps_num);
    while (shape.size() < r.size()) shape.insert(shape.begin(), 1);
    while (r.size() < shape.size()) r.insert(r.begin(), 1);
    const size_t nd = shape.size();
    std::vector<int> inter, bshape, out;
    for (size_t i = 0; i < nd; ++i) {
        OMX_REQUIRE(r[i] >= 0, "mlx_tile: negative repetition");
        inter.push_back(1); inter.push_back(shape[i]);
        bshape.push_back(r[i]); bshape.push_back(shape[i]);
        out.push_back(r[i] * shape[i]);
    }
    Tmp v, bc;
    if (mlx_reshape(&v, arr, inter.data(), inter.size(), s) || mlx_broadcast_to(&bc, v, bshape.data(), bshape.size(), s)) return 1;
    return mlx_reshape(res, bc, out.data(), out.size(), s);
}

// the diagonal as a strided view (element stride = the two axes' strides together)
int mlx_diagonal(mlx_array* res, const mlx_array a, int offset, int axis1, int axis2, const mlx_stream) {
    REQ_ARR(a, "mlx_diagonal");
    const Arr& s = *A(a);
    const int nd = (int)s.shape.size();
    int a1, a2;
    if (norm_axis(axis1, nd, "mlx_diagonal", &a1) || norm_axis(axis2, nd, "mlx_diagonal", &a2)) return 1;
    OMX_REQUIRE(a1 != a2, "mlx_diagonal: the two axes must differ");
    const int n1 = s.shape[a1], n2 = s.shape[a2];
    const int len = offset >= 0 ? std::max(0, std::min(n1, n2 - offset)) : std::max(0, std::min(n1 + offset, n2));
    Arr* r = new Arr(s);
    r->host.clear();
    r->shape.clear(); r->strides.clear();
    for (int i = 0; i < nd; ++i)
        if (i != a1 && i != a2) { r->shape.push_back(s.shape[i]); r->strides.push_back(s.strides[i]); }
    r->shape.push_back(len);
    r->strides.push_back(s.strides[a1] + s.strides[a2]);
    if (len > 0) r->off += (offset >= 0 ? (size_t)offset * s.strides[a2] : (size_t)(-offset) * s.strides[a1]) * dsize(s.dt);
    return assign(res, r);
}
int mlx_diag(mlx_array* res, const mlx_array a, int k, const mlx_stream s) {
    REQ_ARR(a, "mlx_diag");
    const Arr& x = *A(a);
    if (x.shape.size() == 2) return mlx_diagonal(res, a, k, 0, 1, s);
    OMX_REQUIRE(x.shape.size() == 1, "mlx_diag: a vector or a matrix expected");
    const int L = x.shape[0], ak = k < 0 ? -k : k, n = L + ak;
    // row i of the result carries a[i] at column i + k (k >= 0); column j carries a[j] at row j - k (k < 0)
    Tmp zero_pad, padded, vec2, mask, zero, z;
    const int axes[1] = {0}, lo[1] = {0}, hi[1] = {ak};
    zero.a = f32_scalar(0.f);
    if (mlx_astype(&z, zero, x.dt, s) || mlx_pad(&padded, a, axes, 1, lo, 1, hi, 1, z, "constant", s)) return 1;
    const int col[2] = {n, 1}, row[2] = {1, n};
    if (mlx_reshape(&vec2, padded, k >= 0 ? col : row, 2, s) || index_mask(&mask, n, n, k, true, s)) return 1;
    return mlx_where(res, mask, vec2, z, s);
}

int mlx_nan_to_num(mlx_array* res, const mlx_array a, float nan, mlx_optional_float posinf, mlx_optional_float neginf, const mlx_stream s) {
    REQ_ARR(a, "mlx_nan_to_num");
    const mlx_dtype dt = A(a)->dt;
    if (!is_float(dt)) return mlx_array_set(res, a);
    const float big = dt == MLX_FLOAT16 ? 65504.0f : dt == MLX_BFLOAT16 ? 3.3895313892515355e38f : 3.4028234663852886e38f;
    Tmp vn32, vp32, vm32, vn, vp, vm, isn, isp, ism, t1, t2;
    vn32.a = f32_scalar(nan); vp32.a = f32_scalar(posinf.has_value ? posinf.value : big); vm32.a = f32_scalar(neginf.has_value ? neginf.value : -big);
    if (mlx_astype(&vn, vn32, dt, s) || mlx_astype(&vp, vp32, dt, s) || mlx_astype(&vm, vm32, dt, s)) return 1;
    if (mlx_isnan(&isn, a, s) || mlx_isposinf(&isp, a, s) || mlx_isneginf(&ism, a, s)) return 1;
    if (mlx_where(&t1, isn, vn, a, s) || mlx_where(&t2, isp, vp, t1, s)) return 1;
    return mlx_where(res, ism, vm, t2, s);
}

int mlx_broadcast_arrays(mlx_vector_array* res, const mlx_vector_array inputs, const mlx_stream s) {
    OMX_REQUIRE(res && inputs.ctx, "mlx_broadcast_arrays: empty handle");
    const std::vector<Arr*>& in = reinterpret_cast<Vec*>(inputs.ctx)->v;
    size_t nd = 0;
    for (const Arr* a : in) nd = std::max(nd, a->shape.size());
    std::vector<int> shape(nd, 1);
    for (const Arr* a : in)
        for (size_t i = 0; i < a->shape.size(); ++i) {
            const size_t at = nd - a->shape.size() + i;
            const int d = a->shape[i];
            OMX_REQUIRE(d == 1 || shape[at] == 1 || shape[at] == d, "mlx_broadcast_arrays: shapes are not broadcastable (%d vs %d)", d, shape[at]);
            if (d != 1) shape[at] = d;
        }
    Vec* out = new Vec();
    for (Arr* a : in) {
        mlx_array h{a};
        Tmp b;
        if (mlx_broadcast_to(&b, h, shape.data(), shape.size(), s)) { for (Arr* p : out->v) delete p; delete out; return 1; }
        Arr* c = new Arr(*A(b.a));
        c->host.clear();
        out->v.push_back(c);
    }
    if (res->ctx != inputs.ctx) delete reinterpret_cast<Vec*>(res->ctx);
    res->ctx = out;
    return 0;
}


// ---- a few more views / accessors / products ----
const bool* mlx_array_data_bool(const mlx_array arr) { return (const bool*)data_host(arr); }
const int8_t* mlx_array_data_int8(const mlx_array arr) { return (const int8_t*)data_host(arr); }
const int16_t* mlx_array_data_int16(const mlx_array arr) { return (const int16_t*)data_host(arr); }
const int64_t* mlx_array_data_int64(const mlx_array arr) { return (const int64_t*)data_host(arr); }
const uint64_t* mlx_array_data_uint64(const mlx_array arr) { return (const uint64_t*)data_host(arr); }

// an arbitrary strided window onto the array's buffer (strides and offset in elements); it must stay inside the buffer
int mlx_as_strided(mlx_array* res, const mlx_array a, const int* shape, size_t shape_num, const int64_t* strides, size_t strides_num, size_t offset,
                   const mlx_stream) {
    REQ_ARR(a, "mlx_as_strided");
    OMX_REQUIRE(shape_num == strides_num, "mlx_as_strided: one stride per dimension");
    const Arr& s = *A(a);
    size_t last = offset;
    bool empty = false;
    for (size_t i = 0; i < shape_num; ++i) {
        OMX_REQUIRE(shape[i] >= 0 && strides[i] >= 0, "mlx_as_strided: negative shape or stride");
        if (shape[i] == 0) empty = true;
        else last += (size_t)(shape[i] - 1) * (size_t)strides[i];
    }
    OMX_REQUIRE(empty || s.off + (last + 1) * dsize(s.dt) <= s.buf->bytes, "mlx_as_strided: the window leaves the array's buffer");
    Arr* r = new Arr(s);
    r->host.clear();
    r->shape.assign(shape, shape + shape_num);
    r->strides.assign(strides, strides + strides_num);
    r->off = s.off + offset * dsize(s.dt);
    return assign(res, r);
}
// reinterpret the bytes as another dtype; a different item size rescales the last (contiguous) axis
int mlx_view(mlx_array* res, const mlx_array a, mlx_dtype dtype, const mlx_stream) {
    REQ_ARR(a, "mlx_view");
    Contig c;
    if (c.init(*A(a))) return 1;
    const size_t from = dsize(c.a->dt), to = dsize(dtype);
    std::vector<int> shape = c.a->shape;
    if (from != to) {
        OMX_REQUIRE(!shape.empty() && ((size_t)shape.back() * from) % to == 0, "mlx_view: the last axis does not hold a whole number of the new items");
        shape.back() = (int)((size_t)shape.back() * from / to);
    }
    Arr* r = new Arr(*c.a);
    r->host.clear();
    r->dt = dtype;
    r->shape = shape;
    r->strides = row_major(shape);
    return assign(res, r);
}
int mlx_real(mlx_array* res, const mlx_array a, const mlx_stream) { REQ_ARR(a, "mlx_real"); return mlx_array_set(res, a); }   // (no complex dtype here)
int mlx_imag(mlx_array* res, const mlx_array a, const mlx_stream s) {
    REQ_ARR(a, "mlx_imag");
    return mlx_zeros(res, A(a)->shape.data(), A(a)->shape.size(), A(a)->dt, s);
}

// contraction of the listed axes: they move to the end of a and the front of b, the rest is one matrix product
int mlx_tensordot(mlx_array* res, const mlx_array a, const mlx_array b, const int* axes_a, size_t axes_a_num, const int* axes_b, size_t axes_b_num,
                  const mlx_stream s) {
    REQ_ARR(a, "mlx_tensordot"); REQ_ARR(b, "mlx_tensordot");
    OMX_REQUIRE(axes_a_num == axes_b_num, "mlx_tensordot: as many axes of a as of b");
    const Arr &x = *A(a), &y = *A(b);
    const int na = (int)x.shape.size(), nb = (int)y.shape.size();
    std::vector<bool> ca((size_t)na, false), cb((size_t)nb, false);
    std::vector<int> la, lb;
    long long K = 1;
    for (size_t i = 0; i < axes_a_num; ++i) {
        int p, q;
        if (norm_axis(axes_a[i], na, "mlx_tensordot", &p) || norm_axis(axes_b[i], nb, "mlx_tensordot", &q)) return 1;
        OMX_REQUIRE(!ca[p] && !cb[q] && x.shape[p] == y.shape[q], "mlx_tensordot: axes %d / %d do not match (%d vs %d) or repeat", axes_a[i], axes_b[i], x.shape[p], y.shape[q]);
        ca[p] = true; cb[q] = true; la.push_back(p); lb.push_back(q);
        K *= x.shape[p];
    }
    std::vector<int> pa, pb, out_shape;
    long long Ma = 1, Nb = 1;
    for (int i = 0; i < na; ++i) if (!ca[i]) { pa.push_back(i); out_shape.push_back(x.shape[i]); Ma *= x.shape[i]; }
    pa.insert(pa.end(), la.begin(), la.end());
    pb = lb;
    for (int i = 0; i < nb; ++i) if (!cb[i]) { pb.push_back(i); out_shape.push_back(y.shape[i]); Nb *= y.shape[i]; }
    Tmp ta, tb, ma, mb, prod;
    const int sa[2] = {(int)Ma, (int)K}, sb[2] = {(int)K, (int)Nb};
    if (mlx_transpose_axes(&ta, a, pa.data(), pa.size(), s) || mlx_transpose_axes(&tb, b, pb.data(), pb.size(), s) ||
        mlx_reshape(&ma, ta, sa, 2, s) || mlx_reshape(&mb, tb, sb, 2, s) || mlx_matmul(&prod, ma, mb, s))
        return 1;
    return mlx_reshape(res, prod, out_shape.data(), out_shape.size(), s);
}
int mlx_tensordot_axis(mlx_array* res, const mlx_array a, const mlx_array b, int axis, const mlx_stream s) {
    REQ_ARR(a, "mlx_tensordot_axis"); REQ_ARR(b, "mlx_tensordot_axis");
    const int na = (int)A(a)->shape.size();
    OMX_REQUIRE(axis >= 0 && axis <= na && axis <= (int)A(b)->shape.size(), "mlx_tensordot_axis: %d axes cannot be contracted", axis);
    std::vector<int> aa, bb;
    for (int i = 0; i < axis; ++i) { aa.push_back(na - axis + i); bb.push_back(i); }
    return mlx_tensordot(res, a, b, aa.data(), aa.size(), bb.data(), bb.size(), s);
}
// Kronecker product of two matrices (or vectors): a[i, j] * b[k, l] at [i * rows(b) + k, j * cols(b) + l]
int mlx_kron(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s) {
    REQ_ARR(a, "mlx_kron"); REQ_ARR(b, "mlx_kron");
    const Arr &x = *A(a), &y = *A(b);
    OMX_REQUIRE(x.shape.size() <= 2 && y.shape.size() <= 2 && x.shape.size() == y.shape.size() && !x.shape.empty(), "mlx_kron: two vectors or two matrices expected");
    Tmp xa, yb, prod;
    if (x.shape.size() == 1) {
        const int s1[2] = {x.shape[0], 1}, s2[2] = {1, y.shape[0]}, so[1] = {x.shape[0] * y.shape[0]};
        if (mlx_reshape(&xa, a, s1, 2, s) || mlx_reshape(&yb, b, s2, 2, s) || mlx_multiply(&prod, xa, yb, s)) return 1;
        return mlx_reshape(res, prod, so, 1, s);
    }
    const int s1[4] = {x.shape[0], 1, x.shape[1], 1}, s2[4] = {1, y.shape[0], 1, y.shape[1]}, so[2] = {x.shape[0] * y.shape[0], x.shape[1] * y.shape[1]};
    if (mlx_reshape(&xa, a, s1, 4, s) || mlx_reshape(&yb, b, s2, 4, s) || mlx_multiply(&prod, xa, yb, s)) return 1;
    return mlx_reshape(res, prod, so, 2, s);
}

// bernoulli(p) = uniform[0, 1) < p with MLX's keyed uniform generator (mlx random.cpp: `uniform(shape, key) < p`)
int mlx_random_bernoulli(mlx_array* res, const mlx_array p, const int* shape, size_t shape_num, const mlx_array key, const mlx_stream s) {
    REQ_ARR(p, "mlx_random_bernoulli");
    Tmp lo, hi, u;
    lo.a = f32_scalar(0.f); hi.a = f32_scalar(1.f);
    if (mlx_random_uniform(&u, lo, hi, shape, shape_num, MLX_FLOAT32, key, s)) return 1;
    return mlx_less(res, u, p, s);
}

// nn::Conv2d (mlx-rs/src/nn/convolution.rs -> ops::conv2d; the FLUX autoencoder's only convolution, flux-klein-mlx/src/autoencoder.rs:110-131).
// bfloat16 fast routes: 1x1 / stride 1 -> the GEMM kernels on [B H W, Cin]; 3x3 / stride 1 / padding 1 where the implicit-GEMM form applies
// (gemm.hpp conv3x3_implicit_supported) -> one zero-bordered copy + one launch per image; everything else the direct kernel above.
int mlx_conv2d(mlx_array* res, const mlx_array input, const mlx_array weight, int stride_0, int stride_1, int padding_0, int padding_1,
               int dilation_0, int dilation_1, int groups, const mlx_stream s) {
    REQ_ARR(input, "mlx_conv2d"); REQ_ARR(weight, "mlx_conv2d");
    Contig cx, cw;
    if (cx.init(*A(input)) || cw.init(*A(weight))) return 1;
    OMX_REQUIRE(cx.a->shape.size() == 4 && cw.a->shape.size() == 4, "mlx_conv2d: input [B, H, W, C_in] and weight [C_out, kH, kW, C_in / groups] expected");
    OMX_REQUIRE(cx.a->dt == cw.a->dt && is_float(cx.a->dt), "mlx_conv2d: input and weight must share a floating dtype");
    Conv2dGeom g = {cx.a->shape[0], cx.a->shape[1], cx.a->shape[2], cx.a->shape[3], 0, 0, cw.a->shape[0], cw.a->shape[1], cw.a->shape[2],
                    stride_0, stride_1, padding_0, padding_1, dilation_0, dilation_1, groups};
    OMX_REQUIRE(stride_0 >= 1 && stride_1 >= 1 && dilation_0 >= 1 && dilation_1 >= 1 && padding_0 >= 0 && padding_1 >= 0 && groups >= 1 &&
                    g.Cin % groups == 0 && g.Cout % groups == 0 && cw.a->shape[3] == g.Cin / groups,
                "mlx_conv2d: bad stride / dilation / padding / groups (C_in %d, C_out %d, groups %d, weight C_in %d)", g.Cin, g.Cout, groups, cw.a->shape[3]);
    const int span0 = dilation_0 * (g.kH - 1) + 1, span1 = dilation_1 * (g.kW - 1) + 1;
    OMX_REQUIRE(g.H + 2 * padding_0 >= span0 && g.W + 2 * padding_1 >= span1, "mlx_conv2d: kernel span %d x %d exceeds the padded input %d x %d", span0, span1,
                g.H + 2 * padding_0, g.W + 2 * padding_1);
    g.Ho = (g.H + 2 * padding_0 - span0) / stride_0 + 1;
    g.Wo = (g.W + 2 * padding_1 - span1) / stride_1 + 1;
    std::vector<int> shape = {g.B, g.Ho, g.Wo, g.Cout};
    const bool bf16 = cx.a->dt == MLX_BFLOAT16, unit = stride_0 == 1 && stride_1 == 1 && groups == 1;
    if (bf16 && unit && g.kH == 3 && g.kW == 3 && padding_0 == 1 && padding_1 == 1 && dilation_0 == 1 && dilation_1 == 1 && g.B > 0 &&
        omx::conv3x3_implicit_supported(g.H, g.W, g.Cin, g.Cout)) {
        const int axes[2] = {1, 2}, one[2] = {1, 1};
        Tmp zero, padded;
        zero.a = f32_scalar(0.f);
        if (mlx_pad(&padded, input, axes, 2, one, 2, one, 2, zero, "constant", s)) return 1;
        Contig cp;
        if (cp.init(*A(padded))) return 1;
        NEW_OR_FAIL(r, shape, MLX_BFLOAT16);
        for (int b = 0; b < g.B; ++b)
            if (omx::launch_conv3x3_implicit((omx::bf16_t*)r->ptr() + (size_t)b * g.H * g.W * g.Cout,
                                             (const omx::bf16_t*)cp.a->ptr() + (size_t)b * (g.H + 2) * (g.W + 2) * g.Cin, (const omx::bf16_t*)cw.a->ptr(), nullptr,
                                             nullptr, g.H, g.W, g.Cin, g.Cout, g_stream)) {
                delete r;
                return 1;
            }
        return assign(res, r);
    }
    NEW_OR_FAIL(r, shape, cx.a->dt);
    if (r->size()) {
        const long long M = (long long)g.B * g.H * g.W;
        if (bf16 && unit && g.kH == 1 && g.kW == 1 && padding_0 == 0 && padding_1 == 0 && g.Cin % 8 == 0 && M <= 0x7FFFFFFF) {
            if (omx::launch_gemm_bf16((omx::bf16_t*)r->ptr(), (const omx::bf16_t*)cx.a->ptr(), (const omx::bf16_t*)cw.a->ptr(), nullptr, (int)M, g.Cout, g.Cin, g_stream)) {
                delete r;
                return 1;
            }
        } else {
            conv2d_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), cx.a->ptr(), cw.a->ptr(), cx.a->dt, g);
            OMX_LAUNCH_CHECK();
        }
    }
    return assign(res, r);
}

}  // extern "C"
