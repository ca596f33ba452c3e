/*
 * svgr.h -- C ABI of libsvgr_hip.so, the MI355X (gfx950) anti-aliased path rasterizer.
 *
 * The reference (aslpavel/svgrasterize.py) has no FFI layer: its boundary for this path is three
 * Python call signatures plus the Layer value type.  Each entry point below names the
 * reference interface it stands behind ("S:n" = svgrasterize.py line n).  The Python host classes in
 * svgrasterize.py_amd/ bind these with ctypes; INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative svgr_status; svgr_last_error() gives the text
 *     (thread local).  No C++ exception crosses the boundary.
 *   - plain pointers and sizes only.  Host pointers are caller-owned; device memory is an opaque
 *     svgr_buf handle (or a raw device pointer wrapped with svgr_buf_wrap, e.g. torch's data_ptr()).
 *   - one svgr_ctx per device; calls on one context are serialised by the caller; all kernels of a
 *     context run on its HIP stream; functions that return host-visible data synchronise that stream.
 *   - coordinates follow the reference: a point is (row, col) in presentation space; a bbox /
 *     viewport is {row0, col0, rows, cols} (S:88-89, S:966-975).
 *   - all pixel arithmetic is IEEE double with the reference's operation order; the canvas can be
 *     stored as float32 (the contract of BASELINE.json: within 1 ULP of float32(reference)) or double.
 */
#ifndef SVGR_H
#define SVGR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: svgr_batch_set_groups, svgr_batch_set_gradients, svgr_batch_plan_many, svgr_batch_render_window added; svgr_gradient.n_stops
 *    no longer capped at 32; SVGR_RENDER_DETERMINISTIC
 * 3: SVGR_RENDER_SAME_GEOMETRY, SVGR_OUT_FILLS_F64, svgr_layer_convert_to, svgr_layer_scale_to, svgr_batch_render_windows added
 *    (nothing changed or removed)
 * 4: svgr_hash_buffers added (nothing changed or removed)
 * 6: svgr_batch_draw, svgr_measure_begin / _end / _launches added (nothing changed or removed)
 * 5: svgr_layer_compose_over / _in, svgr_layer_convert_scale_to, svgr_layer_convolve_ops, svgr_batch_get_extents added (nothing changed
 *    or removed) */
#define SVGR_ABI_VERSION 6

typedef enum {
    SVGR_OK = 0,
    SVGR_E_INVALID = -1,   /* bad argument (reference raises ValueError, S:945, S:989, S:298) */
    SVGR_E_HIP = -2,       /* HIP runtime error */
    SVGR_E_NOMEM = -3,
    SVGR_E_NODEVICE = -4,  /* no gfx950 device visible */
    SVGR_E_OVERFLOW = -5,  /* internal capacity exceeded (re-plan) or flatten depth cap hit */
    SVGR_E_STATE = -6      /* call order (render before plan, ...) */
} svgr_status;

typedef struct svgr_ctx svgr_ctx;
typedef struct svgr_buf svgr_buf;
typedef struct svgr_batch svgr_batch;

/* segment kinds in svgr_batch_desc.seg_kind */
#define SVGR_SEG_LINE 0   /* PATH_LINE / PATH_CLOSED / PATH_UNCLOSED (S:865-873): 2 points */
#define SVGR_SEG_CUBIC 1  /* PATH_CUBIC, and PATH_QUAD / PATH_ARC after host conversion: 4 points */

/* fill rules (S:874-875, S:984-989) */
#define SVGR_FILL_NONZERO 0
#define SVGR_FILL_EVENODD 1
/* optional flags or-ed into path_rule (canvas outputs only): a CLIP node whose clip and target are single paths
 * (Scene.render RENDER_CLIP, S:698-715: `Layer.compose([mask, image], COMPOSE_IN)`) is two consecutive paths:   */
#define SVGR_PATH_CLIP_SOURCE 2 /* coverage only (Path.mask of the clip path); not painted                        */
#define SVGR_PATH_CLIPPED 4     /* fill multiplied by the coverage of the PREVIOUS path, which must be a clip source */

/* output kinds of svgr_batch_render */
#define SVGR_OUT_CANVAS_F32 0  /* (rows, cols, 4) float32, all paths composited OVER in paint order */
#define SVGR_OUT_CANVAS_F64 1  /* same in double (Layer.image dtype of the reference, S:42) */
#define SVGR_OUT_MASK_F64 2    /* single path: (rows, cols) double coverage = Path.mask().image[..., 0] */
#define SVGR_OUT_FILL_F64 3    /* single path: (rows, cols, 4) double = mask * paint (S:1019) */
#define SVGR_OUT_MASKS_F64 4   /* every path of the batch: its Path.mask (rows_p, cols_p) double, back to back in path order;
                                * layer p starts at sum_{q<p} rows_q*cols_q doubles (clipped bboxes of svgr_batch_get_bboxes,
                                * empty ones count 0).  One launch for all the masks a scene needs (clips, gradient fills). */
#define SVGR_OUT_FILLS_F64 5   /* every path of the batch: its Path.fill layer (rows_p, cols_p, 4) double = mask * paint, back to back in
                                * path order (layer p starts at 4 * sum_{q<p} rows_q*cols_q doubles): the solid fills a document
                                * draws node by node (children of filter / mask nodes) in one launch. */

/* flags of svgr_batch_render */
#define SVGR_RENDER_CLIP01 1u  /* clip RGBA to [0, 1] on store (canvas_merge_at, S:326) */
#define SVGR_RENDER_TIMED 2u   /* bracket the stages with HIP events (svgr_batch_timings) */
/* Bit-reproducible output: the reference's np.cumsum (S:983) adds a row's pieces in one fixed order; the default render
 * adds them in whatever order the LDS atomics of several waves land (results differ in the last bits of a double, i.e.
 * in float32 rounding ties).  With this flag one wave per workgroup does every accumulation, in list order: two
 * renders of the same batch are bit-identical, at roughly half the speed of the geometry and scatter phases. */
#define SVGR_RENDER_DETERMINISTIC 4u
/* svgr_batch_render_window only: this window belongs to the same picture as the window the previous render of this batch drew
 * -- no input of the batch has changed in between --, so the geometry kernels' results are taken as they are instead of being
 * produced again (a document whose runs of fills share one batch draws each run's window from ONE geometry pass).  Ignored
 * (the pass runs) after any svgr_batch_set_* / plan since that render. */
#define SVGR_RENDER_SAME_GEOMETRY 8u

/* -------------------------------------------------------------------------------------------- */
/* context + device memory                                                                      */
/* -------------------------------------------------------------------------------------------- */
int svgr_abi_version(void);
int svgr_tile_rows(void); /* band height: the row granularity of svgr_batch_set_bands */
int svgr_tile_cols(void);
const char* svgr_last_error(void);
int svgr_device_count(void);

int svgr_init(int device_id, svgr_ctx** out);
int svgr_shutdown(svgr_ctx* ctx);
/* run the context's kernels on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int svgr_set_stream(svgr_ctx* ctx, void* hip_stream);
int svgr_sync(svgr_ctx* ctx);
int svgr_device_name(svgr_ctx* ctx, char* out, size_t cap);
/* Measurement helpers (bench.py; the reference prints one wall-clock figure per render, S:3854-3864 -- these split it).
 * svgr_measure_begin holds the context's stream busy for `hold_ms` (0: not at all) and marks the start behind the hold;
 * svgr_measure_end marks the end, waits for it and returns the milliseconds of DEVICE time between the marks: what the caller
 * enqueued in between has queued up behind the hold and run back to back, however long the host took to issue it (less than the
 * hold).  svgr_measure_launches: kernel launches the library has made in this process so far. */
int svgr_measure_begin(svgr_ctx* ctx, double hold_ms);
int svgr_measure_end(svgr_ctx* ctx, double* ms);
int svgr_measure_launches(uint64_t* out);
/* Host only, no GPU involved: a 64-bit hash over the bytes of `n` host buffers (ptrs[i], nbytes[i]), in order.  What the
 * caller's retained renders guard themselves with: the reference's Scene.render (S:649-752) keeps nothing between calls, so a
 * paint or a segment array edited in place is simply drawn with its new values; a caller that keeps built batches between
 * renders of one document checks this hash of the document's arrays before it reuses them. */
int svgr_hash_buffers(const void* const* ptrs, const int64_t* nbytes, int64_t n, uint64_t* out);

int svgr_buf_alloc(svgr_ctx* ctx, size_t bytes, svgr_buf** out);
int svgr_buf_wrap(svgr_ctx* ctx, void* device_ptr, size_t bytes, svgr_buf** out); /* non-owning */
int svgr_buf_free(svgr_ctx* ctx, svgr_buf* buf);
void* svgr_buf_ptr(const svgr_buf* buf);
size_t svgr_buf_bytes(const svgr_buf* buf);
int svgr_buf_zero(svgr_ctx* ctx, svgr_buf* buf);
int svgr_buf_copy(svgr_ctx* ctx, svgr_buf* dst, const svgr_buf* src, size_t bytes); /* device to device, async */
int svgr_upload(svgr_ctx* ctx, svgr_buf* dst, size_t dst_off, const void* host, size_t bytes);
int svgr_download(svgr_ctx* ctx, const svgr_buf* src, size_t src_off, void* host, size_t bytes); /* syncs */

/* -------------------------------------------------------------------------------------------- */
/* batched paths -> coverage -> paint -> composite                                              */
/*                                                                                              */
/* Stands behind Path.mask (S:922-993), Path.fill solid branch (S:995-1019) and the OVER merge  */
/* of a group of fills (Layer.compose -> canvas_merge_union(full=False), S:177-207, S:366-377):  */
/* rendering paths 0..n-1 into the canvas is the same per-pixel operation sequence as composing */
/* their fill layers in painter's order.                                                        */
/* -------------------------------------------------------------------------------------------- */
typedef struct {
    /* geometry in USER space; the device applies path_m6 in the reference's fma form (S:531-534) */
    const double* segs;          /* n_segs x 8: 4 points (x, y); lines use the first two          */
    const uint8_t* seg_kind;     /* n_segs: SVGR_SEG_*                                            */
    int64_t n_segs;
    const int64_t* path_seg_off; /* n_paths + 1 offsets into segs, paint order                    */
    int64_t n_paths;
    const double* path_m6;       /* n_paths x 6: rows 0-1 of the 3x3 transform {m00,m01,m02,m10,m11,m12} */
    const uint8_t* path_rule;    /* n_paths: SVGR_FILL_* | SVGR_PATH_* flags                       */
    const double* path_paint;    /* n_paths x 4 premultiplied RGBA already in the compositing space
                                    (the host applies S:1015-1018 to the 4-vector), times opacity   */
    int64_t viewport[4];         /* {row0, col0, rows, cols}; rows <= 0 means "no viewport"
                                    (only valid for the single-path outputs)                      */
    double flatness;             /* 0.1 in the reference (S:955)                                   */
} svgr_batch_desc;

/* copies the description to HBM; nothing is rendered yet */
int svgr_batch_create(svgr_ctx* ctx, const svgr_batch_desc* desc, svgr_batch** out);
int svgr_batch_destroy(svgr_batch* batch);

/* replace per-path paints / transforms of an existing batch (same counts) */
int svgr_batch_set_paints(svgr_batch* batch, const double* path_paint);
int svgr_batch_set_transforms(svgr_batch* batch, const double* path_m6);

/* Isolated groups inside one batch (Scene.render CLIP / OPACITY over a GROUP of solid fills, S:674-715; replaces one
 * Scene.render recursion + Layer.compose round trip per group): path_group[p] = the group path p belongs to, or -1; the
 * members of a group are consecutive paths.  When a group closes it is, as a whole, multiplied by the coverage of
 * group_clip_src[g] (a SVGR_PATH_CLIP_SOURCE path right in front of the group's first member; -1: no clip:
 * `Layer.compose([mask, group], COMPOSE_IN)`, S:712) and by group_opacity[g] (`Layer.opacity`, S:171-175; 1: none), then
 * composited OVER what lies under it.  Canvas outputs only; n_groups = 0 removes the groups.  Call before svgr_batch_plan. */
int svgr_batch_set_groups(svgr_batch* batch, const int32_t* path_group, int64_t n_groups, const int32_t* group_clip_src,
                          const double* group_opacity);

/* Multi-GPU sharding: rank `rank` of `world` keeps the strips s with s % world == rank, a strip being
 * `strip_bands` consecutive bands (band = svgr_tile_rows() scanlines).  The rank still computes every path's
 * exact bbox, but only flattens-to-memory, bins and renders what reaches its own bands; its output buffer holds
 * the owned bands packed in increasing order.  Default (0, 1, 1) = everything.  Invalidates the plan.       */
int svgr_batch_set_bands(svgr_batch* batch, int rank, int world, int strip_bands);

/* Geometry pass with host read-backs: flatten, per-path bbox, band binning; sizes every work
 * buffer.  Must run once before svgr_batch_render and again after geometry/viewport changes.
 * Synchronises.                                                                                 */
int svgr_batch_plan(svgr_batch* batch);

/* svgr_batch_plan for many batches behind one wait per stream: a document's per-node route plans dozens of small batches
 * (Scene.render: one per run of fills between two filter nodes, S:674-688), and a plan alone is a device round trip whose
 * kernels are a fraction of it.  Same result and same errors as calling svgr_batch_plan on each (the first error ends it). */
int svgr_batch_plan_many(svgr_batch** batches, int64_t n);

typedef struct {
    int64_t n_edges;        /* flattened edges E                                                 */
    int64_t path_pixels;    /* P = sum over paths of clipped bbox rows*cols (SURVEY 8d unit)      */
    int64_t n_band_segs;    /* edge x band records                                               */
    int64_t n_path_bands;   /* (path, band) pairs                                                */
    int64_t n_nonempty;     /* paths with a non-empty clipped bbox                                */
    int64_t bbox_union[4];  /* union of the non-empty bboxes {row0, col0, rows, cols}             */
    int64_t tile_rows, tile_cols;
} svgr_batch_stats;
int svgr_batch_get_stats(const svgr_batch* batch, svgr_batch_stats* out);
/* per-path clipped integer bbox {row0, col0, rows, cols}; rows <= 0 = empty (Path.mask -> None) */
int svgr_batch_get_bboxes(const svgr_batch* batch, int32_t* out /* n_paths x 4 */);
/* per-path extent of ALL its flattened points, unclipped, in presentation space: {min row, min col, max row, max col} doubles --
 * the bounding box of ConvexHull(lines) (S:993, S:2010-2020) under a transform that keeps the axes apart, i.e. the frame of an
 * objectBoundingBox paint (S:1023-1027) without fetching the hull.  Read from the plan's own geometry pass: valid between
 * svgr_batch_plan and the first render (SVGR_E_STATE otherwise).  A path without edges: {+inf, +inf, -inf, -inf}.
 * (svgr_batch_set_gradients with the same path -> gradient assignment and new descriptions -- those frames -- keeps the plan
 * and its pass.)  Defined for the paths this rank KEEPS: under svgr_batch_set_bands with world > 1 a path none of whose rows can
 * reach an owned band is not flattened here and reports {+inf, +inf, -inf, -inf} like a path without edges. */
int svgr_batch_get_extents(svgr_batch* batch, double* out /* n_paths x 4 */);
/* flattened edges in presentation space, (E, 2, 2) doubles, grouped by path; edge_path may be NULL */
int svgr_batch_get_edges(const svgr_batch* batch, double* edges, int32_t* edge_path, int64_t cap);
/* ALL flattened edges of the batch, also those that cannot reach the viewport: the point set of Path.mask's
 * ConvexHull(lines) (S:993), from which objectBoundingBox clips, gradients and patterns take their frame.  Call with
 * edges == NULL to learn the count (*n_edges), then with a buffer of at least that many (2, 2) doubles.          */
int svgr_batch_all_edges(svgr_batch* batch, double* edges, int32_t* edge_path, int64_t cap, int64_t* n_edges);

/* Full device pipeline, asynchronous on the context stream, no host read-back:
 * transform+flatten -> bbox -> band binning -> tile kernel (LDS delta-coverage scatter, row scan,
 * fill rule, paint, source-over) -> store.  `out` must hold the output kind's bytes:
 *   canvas kinds: owned_rows x viewport cols x 4 (owned_rows = all rows unless bands are restricted,
 *   then the owned bands packed in order); single-path kinds: bbox rows x cols (x 4).            */
int svgr_batch_render(svgr_batch* batch, svgr_buf* out, int out_kind, unsigned flags);
int64_t svgr_batch_owned_rows(const svgr_batch* batch);
/* The same for a window of the canvas: `window` = {row0, col0, rows, cols} in presentation pixels, inside the batch's
 * viewport (or, without one, inside the union bbox the plan found); `out` holds rows x cols x 4 of the canvas kind.
 * Only the tiles the window touches are worked on and nothing outside it is written: the layer `Layer.compose` returns for
 * a run of fills covers the union of their bboxes, not the viewport (canvas_merge_union, S:366-379).  The pixels are
 * those svgr_batch_render writes at the same positions, bit for bit.  Canvas outputs, unsharded batches.            */
int svgr_batch_render_window(svgr_batch* batch, svgr_buf* out, int out_kind, unsigned flags, const int32_t* window);
/* n windows of the same picture -- `windows` = n x {row0, col0, rows, cols}, outs[i] holds window i -- from ONE geometry pass,
 * drawn side by side: a window of a few dozen tiles takes as long as its heaviest tile, and a document's runs of fills are
 * dozens of such windows.  Everything enqueued on the context's stream before the call is in front of the windows, everything
 * enqueued after it behind them.  Canvas outputs, unsharded batches; not with SVGR_RENDER_TIMED / _DETERMINISTIC. */
int svgr_batch_render_windows(svgr_batch* batch, int64_t n, svgr_buf* const* outs, int out_kind, unsigned flags,
                              const int32_t* windows);

/* Plan (when the batch has no valid plan: a new batch, or one whose transforms / bands were set since) AND render, behind ONE
 * host wait: a frame with new geometry -- the reference's only mode: every `Path.mask` flattens and rasterises from scratch
 * (S:948-957), `scene.render` is timed as a whole (S:3854-3864).  The tile kernel is enqueued right behind the plan's full
 * geometry pass and the pass is validated when the call's single wait returns; a batch planned before takes ONE flatten
 * traversal.  When a speculative capacity did not hold, svgr_batch_plan + svgr_batch_render run instead (same picture).
 * On return the picture is in `out`, the stream has drained and the batch is planned (svgr_batch_get_stats / _bboxes are valid);
 * a batch that was planned already is simply rendered, then waited for.  Same arguments as svgr_batch_render.           */
int svgr_batch_draw(svgr_batch* batch, svgr_buf* out, int out_kind, unsigned flags);

/* HIP-event timings accumulated over the SVGR_RENDER_TIMED renders since the last call
 * (synchronises).  ms_geometry = transform/flatten/bbox/binning kernels, ms_tile = the tile
 * kernel (the coverage+composite pass).                                                         */
int svgr_batch_timings(svgr_batch* batch, int* n_renders, double* ms_total, double* ms_geometry, double* ms_tile);

/* -------------------------------------------------------------------------------------------- */
/* Layer operations on double device images (Layer.image stays in HBM until read)               */
/* bbox arguments are {row0, col0, rows, cols}; images are (rows, cols, channels) row-major      */
/* -------------------------------------------------------------------------------------------- */
/* canvas_merge_union(full=False) step, S:366-377 + S:286: dst(4ch) = src + dst*(1-src_a) on the
 * overlap; `first` copies instead (S:374-375).  src_channels 1 broadcasts (S:283-286).          */
int svgr_layer_over(svgr_ctx* ctx, svgr_buf* dst, const int64_t* dst_bbox, const svgr_buf* src,
                    const int64_t* src_bbox, int src_channels, int first);
/* canvas_merge_intersect, S:382-416: `out` (4ch, bbox = intersection) initialised from `first`
 * (1ch broadcast or 4ch crop), then svgr_layer_in multiplies: out = src * out_alpha (S:290).    */
int svgr_layer_crop4(svgr_ctx* ctx, svgr_buf* out, const int64_t* out_bbox, const svgr_buf* first,
                     const int64_t* first_bbox, int first_channels);
int svgr_layer_in(svgr_ctx* ctx, svgr_buf* out, const int64_t* out_bbox, const svgr_buf* src,
                  const int64_t* src_bbox, int src_channels);
/* Layer.opacity, S:171-175: image * opacity */
int svgr_layer_scale(svgr_ctx* ctx, svgr_buf* img, int64_t n_values, double factor);
/* ... into another buffer (dst may be src): Layer.opacity returns a new layer, and a copy followed by the in-place form
 * reads and writes the image twice */
int svgr_layer_scale_to(svgr_ctx* ctx, svgr_buf* dst, const svgr_buf* src, int64_t n_values, double factor);
/* the clip(0, 1) that ends canvas_merge_at (S:326), in place on n_values doubles */
int svgr_layer_clip01(svgr_ctx* ctx, svgr_buf* img, int64_t n_values);
/* Layer.background, S:166-169: premultiplied linear RGBA image OVER the constant colour rgba[4], in place */
int svgr_layer_background(svgr_ctx* ctx, svgr_buf* img, int64_t n_px, const double* rgba);
/* Layer.convert, S:129-164 + S:471-503, in place on n_px RGBA pixels.
 * ops bitmask applied in this order: 1 = premultiplied->straight, 2 = sRGB->linear,
 * 4 = linear->sRGB, 8 = straight->premultiplied                                                */
int svgr_layer_convert(svgr_ctx* ctx, svgr_buf* img, int64_t n_px, unsigned ops);
/* ... into another buffer (dst may be src) */
int svgr_layer_convert_to(svgr_ctx* ctx, svgr_buf* dst, const svgr_buf* src, int64_t n_px, unsigned ops);
/* Layer.opacity of a layer that still needs its Layer.convert (S:171-175: `convert(pre_alpha=True, ...)`, then image * opacity):
 * both in one pass over n_px RGBA pixels, dst may be src */
int svgr_layer_convert_scale_to(svgr_ctx* ctx, svgr_buf* dst, const svgr_buf* src, int64_t n_px, unsigned ops, double factor);
/* Layer.compose(layers, COMPOSE_OVER) as ONE pass (S:177-207 -> canvas_merge_union, S:366-379): `out` (4ch, bbox = the union) is
 * written whole -- no clear in front, no pass per layer.  Source i is (rows, cols, channels[i]) at src_bboxes[4 i ..]; the first
 * is copied where it covers a pixel (S:374-375), the others go OVER (S:286); pixels no source covers are zero.  ops[i] (or NULL):
 * the svgr_layer_convert ops source i still needs, applied to its pixels as they are read (4-channel sources only).            */
int svgr_layer_compose_over(svgr_ctx* ctx, svgr_buf* out, const int64_t* out_bbox, int64_t n, svgr_buf* const* srcs,
                            const int64_t* src_bboxes, const int32_t* channels, const uint32_t* ops);
/* Layer.compose(layers, COMPOSE_IN) as ONE pass (canvas_merge_intersect, S:382-416): `out` (4ch, bbox = the intersection) =
 * the first source cropped (1ch broadcast), then every other source times the alpha of what is there (S:290), in order; ops as above */
int svgr_layer_compose_in(svgr_ctx* ctx, svgr_buf* out, const int64_t* out_bbox, int64_t n, svgr_buf* const* srcs,
                          const int64_t* src_bboxes, const int32_t* channels, const uint32_t* ops);
/* float64 -> float32 (optionally clipping to [0,1]) for presentation */
int svgr_layer_to_f32(svgr_ctx* ctx, svgr_buf* dst_f32, const svgr_buf* src_f64, int64_t n_values, int clip01);

/* -------------------------------------------------------------------------------------------- */
/* gradient paint servers and Gaussian blur (config 5)                                          */
/* -------------------------------------------------------------------------------------------- */
/* The other canvas_compose modes (S:287-297) on the full union canvas (canvas_merge_union(full=True), S:348-361):
 * out(4ch, bbox ob) = blend(out, src zero-extended), every pixel of ob.  mode: 1 OUT, 3 ATOP, 4 XOR (the reference's
 * COMPOSE_* codes, S:48-51), 5 = arithmetic with k4 = {k1, k2, k3, k4}: clip(k1*src*dst + k2*src + k3*dst + k4, 0, 1).
 * (OVER and IN have their own entry points above; an unknown mode is SVGR_E_INVALID like the ValueError of S:298.)  */
int svgr_layer_blend(svgr_ctx* ctx, svgr_buf* out, const int64_t* ob, const svgr_buf* src, const int64_t* sb, int src_channels,
                     int mode, const double* k4);
/* Layer.color_matrix (S:95-104) in place on a straight-alpha linear RGBA image: clip(px @ M[:, :4].T + M[:, 4], 0, 1);
 * m20 = 4 x 5 row-major host matrix.                                                                            */
int svgr_layer_color_matrix(svgr_ctx* ctx, svgr_buf* img, int64_t n_px, const double* m20);
/* Layer.morphology (S:120-127, pooling S:419-468): min (is_max 0) / max pooling, window ky rows x kx columns, stride 1,
 * no padding: out is (rows - ky + 1, cols - kx + 1, 4).                                                          */
int svgr_layer_morphology(svgr_ctx* ctx, svgr_buf* out, const svgr_buf* src, int64_t rows, int64_t cols, int64_t ky, int64_t kx,
                          int is_max);
/* Luminance of a straight-alpha RGBA image for RENDER_MASK (S:735): out(n_px doubles) = (rgb . {0.2125, 0.7154, 0.072}) * a */
int svgr_layer_luminance(svgr_ctx* ctx, svgr_buf* out_1ch, const svgr_buf* src_rgba, int64_t n_px);

/* Output stage (canvas_to_png, S:262): dst(n_px x 4 uint8) = round-half-even(src(n_px x 4 double) * 255), the
 * quantisation the reference applies to Layer.convert(pre_alpha=False, linear_rgb=False).image before zlib
 * (S:209-213).  Values outside [0, 255] saturate.  The PNG container itself is written on the host.      */
int svgr_layer_to_rgba8(svgr_ctx* ctx, svgr_buf* dst_u8, const svgr_buf* src_f64, int64_t n_px);

/* Path.fill, gradient branch (S:1021-1047): out(rows, cols, 4) = gradient(pixel centre) * mask(rows, cols),
 * GradLinear.fill S:1553-1563, GradRadial.fill S:1577-1650, grad_spread S:1661-1668, grad_interpolate
 * S:1671-1683.  The host supplies what the reference computes once per fill with numpy: the inverse
 * transforms, vec / vec.vec, the focal-circle scalars and the stops already converted to the target
 * colour space (grad_stops_colorspace, S:1686-1695).                                                */
typedef struct {
    int kind;            /* 1 linear, 2 radial (one circle), 3 radial with a focal circle                    */
    int spread;          /* 0 pad, 1 repeat, 2 reflect                                                       */
    int has_gt;          /* apply gt_m6 (gradientTransform inverse) after user_m6                            */
    int n_stops;         /* >= 1; no cap (lists longer than 32 travel in a device buffer, S:1671-1683)        */
    int excl_enabled;    /* fradius != radius: exclude negative r(t) (S:1642-1644)                           */
    double user_m6[6];   /* pixel centre -> user space: rows 0-1 of transform.invert.m (S:1023-1027)          */
    double gt_m6[6];
    double p0[2], vec[2], vv;                       /* linear: p0, p1 - p0, vec . vec                          */
    double center[2], radius;                       /* radial                                                 */
    double fcenter[2], fradius, cd[2], rd, a, frad_rd, frad2, excl_thresh;   /* focal: S:1619-1624, S:1644      */
    const double* stop_off;                         /* n_stops offsets                                        */
    const double* stop_rgba;                        /* n_stops x 4 premultiplied colours                      */
} svgr_gradient;
int svgr_gradient_fill(svgr_ctx* ctx, const svgr_gradient* g, const svgr_buf* mask, const int64_t* bbox, svgr_buf* out_rgba);
/* Gradient paints inside a batch (canvas outputs): path_grad[p] = index into grads or -1 (solid).  One description per
 * gradient-filled path (user_m6 is the fill's own pixel -> user transform), userSpaceOnUse, at most 32 stops; the tile kernel
 * evaluates it per visible pixel with the code of svgr_gradient_fill and multiplies by the coverage and by path_paint[p]
 * (all ones, or the opacity of an OPACITY node directly above the leaf).  n_grads = 0 removes them.  Call before
 * svgr_batch_plan.  The fills of a document then need no Path.mask + svgr_gradient_fill + Layer.compose round trip each. */
int svgr_batch_set_gradients(svgr_batch* batch, const int32_t* path_grad, int64_t n_grads, const svgr_gradient* grads);
/* GradLinear.fill / GradRadial.fill (S:1553-1563, S:1577-1651): the gradient at n_points caller-supplied coordinates
 * (x, y doubles interleaved; user space, i.e. what Path.fill passes after its own user transform: set user_m6 to the
 * identity), no mask; out_rgba receives n_points x 4 doubles.                                                     */
int svgr_gradient_eval(svgr_ctx* ctx, const svgr_gradient* g, const svgr_buf* points, int64_t n_points, svgr_buf* out_rgba);

/* Path.fill, pattern branch (S:1049-1094): out_rgba = pattern tile looked up per pixel * mask.  `tile` is the
 * (rows, cols, 4) double image of the pattern's scene rendered once by the caller (Scene.render under the fill's
 * transform without translation, S:1051-1064); this call does the per-pixel part, S:1066-1094: pixel centre ->
 * inv_m6 -> np.remainder by the cell -> fwd_m6 -> astype(int) -> minus min_xy -> tile pixel clipped to [0, 1]
 * (0 outside the tile), times the coverage.  SVGR_E_INVALID if an offset leaves the pattern canvas (numpy: IndexError). */
typedef struct svgr_pattern {
    double inv_m6[6];      /* repeat transform inverted: presentation pixels -> pattern space (S:1066-1071)      */
    double fwd_m6[6];      /* repeat transform, translation removed: pattern space -> pixels                       */
    double cell[4];        /* Pattern.x, y, width, height                                                         */
    int64_t min_xy[2];     /* integer minimum over the transformed cell corners (S:1084)                          */
    int64_t pat_shape[2];  /* (w + 1, h + 1) of the pattern canvas (S:1088)                                       */
    int64_t tile_bbox[4];  /* tile layer inside that canvas: x - min_x, y - min_y, rows, cols (S:1089)            */
} svgr_pattern;
int svgr_pattern_fill(svgr_ctx* ctx, const svgr_pattern* pattern, const svgr_buf* tile, const svgr_buf* mask,
                      const int64_t* bbox, svgr_buf* out_rgba);

/* Layer.convolve (S:106-118): full 2-D convolution of a (rows, cols, 4) double image with a host (kw, kh)
 * kernel (blur_kernel, S:1903-1944, is built on the host); out is (rows + kw - 1, cols + kh - 1, 4).     */
int svgr_layer_convolve(svgr_ctx* ctx, svgr_buf* out, const svgr_buf* src, int64_t rows, int64_t cols, const double* kernel,
                        int64_t kw, int64_t kh);
/* ... of a source that still needs its Layer.convert (the filter's `source.convert(pre_alpha=False, linear_rgb=True)`, S:1803):
 * src_ops = the svgr_layer_convert ops, applied to the source's pixels as the first pass reads them */
int svgr_layer_convolve_ops(svgr_ctx* ctx, svgr_buf* out, const svgr_buf* src, int64_t rows, int64_t cols, const double* kernel,
                            int64_t kw, int64_t kh, unsigned src_ops);

/* -------------------------------------------------------------------------------------------- */
/* Path.stroke (S:1105-1180), host side: stroke outline of a path as a fill path.               */
/* Input and output use the reference's segment codes (S:865-872) with 8 doubles per segment    */
/* (points interleaved x, y; unused slots 0) and per-subpath segment counts, the layout of       */
/* Path.from_segments.  Quadratic / arc segments must be converted to cubics by the caller      */
/* (bezier2_to_bezier3 S:2182, arc_to_bezier3 S:2355, exactly as Path.stroke does, S:1133-1140). */
/* Offsetting: line_offset S:2328, bezier3_offset S:2113-2179; joins S:1495-1522 (miter limit 4), */
/* caps S:1466-1492.  Output segments are LINE (2 points), QUAD (3, round joins) or CUBIC (4).   */
/* -------------------------------------------------------------------------------------------- */
#define SVGR_PATH_LINE 0
#define SVGR_PATH_QUAD 1
#define SVGR_PATH_CUBIC 2
#define SVGR_PATH_ARC 3
#define SVGR_PATH_CLOSED 4
#define SVGR_PATH_UNCLOSED 5
#define SVGR_CAP_BUTT 0    /* STROKE_CAP_BUTT (default, S:1468) */
#define SVGR_CAP_ROUND 1
#define SVGR_CAP_SQUARE 2
#define SVGR_JOIN_MITER 0  /* STROKE_JOIN_MITER (default, S:1497) */
#define SVGR_JOIN_ROUND 1
#define SVGR_JOIN_BEVEL 2
typedef struct svgr_stroke_out svgr_stroke_out;
int svgr_path_stroke(const int32_t* seg_types, const double* seg_params, const int32_t* subpath_sizes, int64_t n_subpaths,
                     double width, int linecap, int linejoin, svgr_stroke_out** out);
int svgr_stroke_out_counts(const svgr_stroke_out* s, int64_t* n_segs, int64_t* n_subpaths);
int svgr_stroke_out_copy(const svgr_stroke_out* s, int32_t* seg_types, double* seg_params, int32_t* subpath_sizes);
void svgr_stroke_out_free(svgr_stroke_out* s);

#ifdef __cplusplus
}
#endif
#endif /* SVGR_H */
