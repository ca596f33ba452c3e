// Host build of the per-lane arithmetic in svgrasterize.py_amd/csrc/svgr_core.h, for CPU-side unit
// tests only (tests/test_core_host.py).  This is NOT a CPU fallback of the product: the package never
// loads it; it exists so that the arithmetic every HIP kernel runs per lane can be checked against the
// oracle and the reference fixtures on a machine without a GPU.
#include "../svgrasterize.py_amd/csrc/svgr_core.h"

using namespace svgr;

extern "C" {

void hh_xform(const double* m6, const double* in, long n, double* out) {
    for (long i = 0; i < n; ++i) xform_point(m6, in[2 * i], in[2 * i + 1], out[2 * i], out[2 * i + 1]);
}

double hh_flatness(const double* c) { return cubic_flatness(c); }

// mode 0: recursive (explicit stack), mode 1: stack-free form the GPU uses, mode 2: the GPU's
// 32-lane decomposition (lane j owns the depth-5 node with path bits j) emulated lane by lane
long hh_flatten(const double* cubic, double tol, double* edges, long cap, int mode) {
    const double thr = (tol * tol) * 16.0;
    long n = 0;
    auto emit = [&](double r0, double c0, double r1, double c1) {
        if (n < cap) { edges[4 * n] = r0; edges[4 * n + 1] = c0; edges[4 * n + 2] = r1; edges[4 * n + 3] = c1; }
        ++n;
    };
    if (mode == 0) {
        flatten_cubic(cubic, thr, emit);
    } else if (mode == 1) {
        bool ovf = false;
        flatten_subtree(cubic, thr, kMaxFlattenDepth, emit, ovf);
    } else {
        const int SUB = 5;
        for (int sub = 0; sub < (1 << SUB); ++sub) {
            double node[8];
            for (int i = 0; i < 8; ++i) node[i] = cubic[i];
            int m = 2;
            for (int l = 0; l < SUB; ++l) {
                if (cubic_flatness(node) < thr) { m = (sub & ((1 << (SUB - l)) - 1)) == 0 ? 1 : 0; break; }
                double t[8];
                if ((sub >> (SUB - 1 - l)) & 1) cubic_right(node, t); else cubic_left(node, t);
                for (int i = 0; i < 8; ++i) node[i] = t[i];
            }
            if (m == 1) emit(node[0], node[1], node[6], node[7]);
            else if (m == 2) { bool ovf = false; flatten_subtree(node, thr, kMaxFlattenDepth - SUB, emit, ovf); }
        }
    }
    return n;
}

// one edge (layer-local coordinates) into a (rows, cols) trace, through the exact functions the
// kernels use: edge_setup -> row_step -> row_record -> apply_record, with the reference's clamping
void hh_trace_edge(double* trace, long rows, long cols, const double* e, int use_record) {
    EdgeSetup es = edge_setup(e[0], e[1], e[2], e[3], (int)rows);
    if (!es.valid) return;
    RowState st;
    st.x_next = es.x;
    st.x = es.x;
    st.d = 0.0;
    for (int y = es.y_begin; y < es.y_end; ++y) {
        row_step(st, y, es.p0y, es.p1y, es.dxdy, es.dir);
        double* row = trace + (long)y * cols;
        auto put = [&](int xi, double v) -> bool {
            int c = xi > 0 ? xi : 0;
            if (c >= cols) return false;
            row[c] += v;
            return true;
        };
        if (use_record) {
            RowPieces rp = row_record(st.x, st.x_next, st.d);
            apply_record(rp.x0i, rp.n, rp.v, put);
        } else {
            row_pieces(st.x, st.x_next, st.d, put);
        }
    }
}

double hh_fill(double s, int rule) { return fill_rule(s, rule); }
double hh_fill_raw(double s, int rule) { return rule ? fill_evenodd_raw(s) : fill_nonzero_raw(s); }

void hh_over(double* dst, const double* src) { over_px(dst, src[0], src[1], src[2], src[3]); }

unsigned long long hh_key(double v) { return f64_key(v); }
double hh_unkey(unsigned long long k) { return key_f64(k); }

}  // extern "C"
