/* real3daug_hip.h -- C ABI of libreal3daug_hip.so (MI355X / gfx950).
 *
 * The hot path of Real3D-Aug (ctu-vras/pcl-augmentation): spherical projection, range-image
 * min-reduce, 5x3 closing + hole fill, visibility mask, scene cull, sample select and concat.
 * The reference has no FFI/plugin interface for this path -- it is a set of module-level Python
 * functions called from the `__main__` block of Real3DAug/insertion.py -- so
 * every entry point below cites the reference FUNCTION (or inline block) it replaces.  Paths are
 * relative to the reference root; "SS" = semantic_segmentation/, the object_detection copies are
 * byte-identical one line lower.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every data pointer is a DEVICE pointer (hipMalloc /
 *     a PyTorch-ROCm tensor's data_ptr()) unless the comment says "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All calls are
 *     asynchronous and stream-ordered; none of them allocates, frees or synchronises, so a call
 *     sequence can be captured into a hipGraph.
 *   - The library owns no device memory and keeps no state besides a thread-local error string, caches of device
 *     attributes (resident workgroups per kernel shape) and the placement search's helper streams (r3d_places_release).
 *     It reads no environment variable: every switch is a bit of a descriptor.
 *     Scratch is a caller-provided workspace (size from the *_workspace_bytes queries).
 *   - Return value: R3D_OK or a negative R3D_E_* (bad argument, HIP launch error).  Conditions the
 *     reference reports with `assert` / exceptions while iterating over points are accumulated
 *     as R3D_S_* bits in a caller-provided device word `status` (one per scene) which the host
 *     reads when it chooses; no exception crosses the ABI.
 *   - Layouts are the reference's: "pcl9" is the N x 9 float64 scratch record
 *     [x y z r azimuth elevation intensity label pixel-id] of add_space_for_spherical
 *     (SS Real3DAug/insertion.py:54-64); "pcl5" is N x 5 float64 [x y z intensity label]
 *     (SS Real3DAug/tools/datasets.py:51-56); range images are row-major float64
 *     [num_row][num_column] with 500 = empty depth and label +1 / -1 (insertion.py:98-99).
 */
#ifndef REAL3DAUG_HIP_H
#define REAL3DAUG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3D_VERSION 0x00020005   /* 2.5: r3d_batch_begin_xyz, r3d_host_pack_frames_xyz, r3d_host_read_frames_xyz (12 bytes per point over the link in delta mode);
                                  * r3d_batch_insert_first, r3d_batch_export_alive, r3d_places_release, r3d_places_chunk_ranges_f32; r3d_place_query_t grew
                                  * (32 placement labels; R3D_PQ_SCENE_SLAB / R3D_PQ_ORIG_SLAB and their fields); R3D_PS_BAD_DESCRIPTOR; R3D_E_IO;
                                  * 2.4: R3D_MAX_SAMPLE 65 535; R3D_B_FILE_ORDER, r3d_batch_export_pix, r3d_batch_point_order (clouds in no
                                  * file order are numbered anew internally); r3d_host_write_delta_frames, r3d_host_append_text_files; r3d_batch_debug_counters
                                  * holds 64 values; R3D_S_CHAIN_TIMEOUT now means "a scene's chain was left unfinished" (no slot
                                  * waits for another any more); the workspace of a batch grew (r3d_batch_workspace_bytes);
                                  * 2.3: min_points < 0 (the state a rejected candidate leaves), r3d_batch_adopt_rejected; 2.2: r3d_batch_t.pix / far_pix
                                  * hold (row << 16) | column; r3d_batch_debug_trace is gone; r3d_build_info */

#define R3D_NUMROW 112        /* insertion.py:22 */
#define R3D_NUMCOLUMN 1440    /* insertion.py:23; the pixel id always multiplies by THIS (:116,:127) */

/* return codes */
#define R3D_OK 0
#define R3D_E_ARG (-1)        /* null pointer, negative size, unsupported shape */
#define R3D_E_HIP (-2)        /* a HIP runtime call failed; see r3d_last_error() */
#define R3D_E_WORKSPACE (-3)  /* workspace smaller than the matching *_workspace_bytes() */
#define R3D_E_IO (-4)         /* a host file writer / appender could not write: r3d_last_error() names the file that failed and why
                                 (errno as it was at the failing call); the temporary file is removed */

/* device status bits (OR-ed into *status) */
#define R3D_S_NONFINITE 1         /* NaN/Inf coordinate or a point at the origin (r = 0: z/r is NaN,
                                     the reference's int() raises, insertion.py:104) */
#define R3D_S_ROW_RANGE 2         /* scene row outside [0,num_row): assert insertion.py:110 */
#define R3D_S_COL_RANGE 4         /* column outside [0,num_column): assert insertion.py:112 */
#define R3D_S_SAMPLE_TOO_LARGE 8  /* batched path: sample exceeds R3D_MAX_SAMPLE points */
#define R3D_S_CAPACITY 16         /* batched path: merged cloud / log would exceed its capacity */
#define R3D_S_FAR_OVERFLOW 32     /* batched path: more than R3D_FAR_CAP pixels beyond 500 m */
#define R3D_S_CHAIN_TIMEOUT 128   /* r3d_batch_insert_many: the chain of the scene's slots was left unfinished (a slot never
                                     published; the name is round 2's, when slots waited for each other against a clock) */
#define R3D_S_ORDER_PROMISE 256   /* R3D_B_FILE_ORDER was set for r3d_batch_finish / export_delta / export_rows of a batch whose
                                     r3d_batch_begin had looked at the point order and numbered this scene anew: the scene's
                                     results are not valid (the bit belongs to the batch from one begin to the next) */
#define R3D_S_WINDOW_TOO_LARGE 64 /* batched path: the bit images of the insert's window of the range image plus the
                                     sample's per-point arrays exceed one CU's LDS (a sample that covers more
                                     than ~200 000 pixels) */

#define R3D_B_FILE_ORDER 2048     /* r3d_batch_t.reserved: the clouds come in a LiDAR file order (see r3d_batch_export_pix below) */
#define R3D_B_SLOT_LAUNCHES 65536 /* r3d_batch_t.reserved: r3d_batch_insert_many makes one launch per slot (no chain inside a kernel) */
#define R3D_MAX_SAMPLE 65535      /* points per insert candidate in the batched path (16-bit indices into the sample; round 4:
                                     8 192).  A candidate whose per-point arrays exceed a chain workgroup's LDS goes to
                                     k_insert_big, one that exceeds a whole CU's comes back as R3D_S_WINDOW_TOO_LARGE; the
                                     Python mirror runs a frame flagged with that bit, with R3D_S_SAMPLE_TOO_LARGE or with
                                     R3D_S_FAR_OVERFLOW once more through Level 1, which has none of these limits */
#define R3D_FAR_CAP 1024

int r3d_version(void);
const char *r3d_last_error(void);   /* host string, thread-local, valid until the next failing call */
/* "sources <sha256 of the library's sources as csrc/Makefile hashed them>": which sources this binary was built from */
const char *r3d_build_info(void);

/* ============================================================================================
 * Level 1 -- one cloud at a time, the reference's functions one for one (float64 N x 9 layout).
 * ============================================================================================ */

/* a1  add_space_for_spherical(point_cloud)            SS Real3DAug/insertion.py:54-64
 * pcl5 [n][5] -> pcl9 [n][9], every other column -1. */
int r3d_add_space_for_spherical(const double *pcl5, int64_t n, double *pcl9, void *stream);

/* a2  fill_spherical(point_cloud)                     SS Real3DAug/insertion.py:67-81
 * In place on pcl9: col3 = sqrt(x*x+y*y+z*z) (no FMA contraction), col4 = atan2(y,x)+pi,
 * col5 = acos(z/r).  bounds[0] = max elevation, bounds[1] = min elevation (the reference's return
 * order, :81).  n == 0 is R3D_E_ARG (the reference raises on an empty reduction, :78). */
int r3d_fill_spherical(double *pcl9, int64_t n, double *bounds, int32_t *status, void *stream);

/* a3  geometrical_front_view(point_cloud, num_row, num_column, max_el, min_el, sample)
 *                                                     SS Real3DAug/insertion.py:84-129
 * Reads cols 3..5 of pcl9, writes col 8 = row*R3D_NUMCOLUMN+col for every binned point (skipped
 * points keep their value, :107-108), and fills train/label [num_row][num_column].
 * workspace: r3d_front_view_workspace_bytes(num_row, num_column). */
size_t r3d_front_view_workspace_bytes(int32_t num_row, int32_t num_column);
int r3d_geometrical_front_view(double *pcl9, int64_t n, int32_t num_row, int32_t num_column,
                               double max_el, double min_el, int32_t sample,
                               double *train, double *label,
                               void *workspace, size_t workspace_bytes,
                               int32_t *status, void *stream);
/* The same with the reference's GLOBAL NUMCOLUMN as an argument: the pixel id of col 8 is row*id_columns+col.  The
 * reference's function multiplies by the module global (insertion.py:23, :116, :127), not by its num_column
 * argument; a user who works on another grid edits that global -- here it is passed in.  r3d_geometrical_front_view
 * = this with id_columns = R3D_NUMCOLUMN. */
int r3d_geometrical_front_view_grid(double *pcl9, int64_t n, int32_t num_row, int32_t num_column, int32_t id_columns,
                                    double max_el, double min_el, int32_t sample,
                                    double *train, double *label,
                                    void *workspace, size_t workspace_bytes,
                                    int32_t *status, void *stream);

/* a4  class_closing(original_label)                   SS Real3DAug/tools/closing.py:9-23
 * closed [rows][cols] uint8 in {0,255}: grey closing of clip(label,0,1) with rectangle(5,3)
 * (5 rows x 3 columns), windows clipped at the borders, no azimuth wrap. */
int r3d_class_closing(const double *label, int32_t rows, int32_t cols, uint8_t *closed, void *stream);

/* a5  smooth_out(original_train, original_label)      SS Real3DAug/tools/closing.py:26-62
 * Closed-but-empty pixels get sum(original depth of occupied 5x3 neighbours, drow outer / dcolumn
 * inner) / count and label 1; everything else is copied. */
int r3d_smooth_out(const double *train, const double *label, int32_t rows, int32_t cols,
                   double *train_out, double *label_out, void *stream);

/* a6-a8  the inline visibility / cull / select block  SS Real3DAug/insertion.py:463-482
 * (named occlusion_merge in the Python mirror).  scene9 [n][9], sample9 [m][9] with col 8 filled
 * by a3; scene_train / sample_train are the smoothed range images [rows][cols], cols <=
 * R3D_NUMCOLUMN.  Outputs, each [..][9] float64 with capacity n / m / n rows:
 *   scene_out9 = scene rows whose pixel is not visible, original order       (:472-473)
 *   visible9   = sample rows in visible pixels, by pixel id then sample order (:474-482)
 *   covered9   = removed scene rows, by pixel id then scene order            (:470-471)
 * counts (device int64[3]) = rows written to the three outputs. */
size_t r3d_occlusion_merge_workspace_bytes(int64_t n, int64_t m, int32_t rows, int32_t cols);
int r3d_occlusion_merge(const double *scene9, int64_t n, const double *sample9, int64_t m,
                        const double *scene_train, const double *sample_train,
                        int32_t rows, int32_t cols,
                        double *scene_out9, double *visible9, double *covered9, int64_t *counts,
                        void *workspace, size_t workspace_bytes, void *stream);
/* The same for pixel ids row*id_columns+col (insertion.py:470 divides by the global NUMCOLUMN), cols <= id_columns. */
int r3d_occlusion_merge_grid(const double *scene9, int64_t n, const double *sample9, int64_t m,
                             const double *scene_train, const double *sample_train,
                             int32_t rows, int32_t cols, int32_t id_columns,
                             double *scene_out9, double *visible9, double *covered9, int64_t *counts,
                             void *workspace, size_t workspace_bytes, void *stream);

/* a9  remove_space_for_spherical + the casts of save_data
 *                         SS Real3DAug/tools/datasets.py:72-106, OD tools/datasets.py:76-109
 * pcl9 -> xyzi float32 [n][4] (cols 0,1,2,6) and label uint32 [n] (col 7): the bytes of
 * velodyne/{f}.bin and labels/{f}.label.  `check` (nullable) float32 [n][check_cols], check_cols
 * 5 (SS: x y z i label) or 4 (OD: x y z i): the bytes of check/{f}.bin. */
int r3d_remove_space_for_spherical(const double *pcl9, int64_t n, float *xyzi, uint32_t *label,
                                   float *check, int32_t check_cols, void *stream);

/* ============================================================================================
 * Level 2 -- batched pipeline: B scenes advanced in lock step through K inserts
 * (a10, the per-insert outer loop SS Real3DAug/insertion.py:371-381 with the candidate loop
 * :449-545), everything resident in HBM.  See DESIGN.md for the incremental algorithm and the
 * proof obligations (tests/test_incremental_model.py).
 * ============================================================================================ */
typedef struct r3d_batch {
  int32_t B;             /* scenes in the batch */
  int32_t rows, cols;    /* range image shape (reference: 112 x 1440) */
  int32_t reserved;      /* bit 0: diagnostic, project with the reference formula only (no float32 guess) */
  int64_t cap;           /* point capacity per scene (original points + every insert) */
  int64_t log_cap;       /* inserted-point capacity per scene */
  /* the cloud of scene s lives at index s*cap .. s*cap + n_total[s] */
  float *xyzi;           /* [B*cap][4]  x y z intensity as in velodyne .bin files (caller fills [0,n)) */
  uint32_t *label;       /* [B*cap]     semantic label as in .label files, masked with 0xFFFF */
  int32_t *pix;          /* [B*cap]     pixel of every point under the scene's current bounds: (row << 16) | column */
  int32_t *n_head;       /* [B] float32-exact points at the front of the cloud */
  int32_t *n_total;      /* [B] n_head + live/dead inserted points */
  int32_t *tail_ref;     /* [B*log_cap] log row of cloud point n_head + t */
  /* append-only log of accepted visible points = all_visible_parts (insertion.py:534-545) */
  double *log5;          /* [B*log_cap][5] x y z intensity label, float64 as the sample had them */
  int32_t *log_birth;    /* [B*log_cap] step at which the row was inserted */
  int32_t *n_log;        /* [B] */
  /* per-scene range-image state (pixel ids above; liveness, chunk boxes and tile counts live in the workspace) */
  double *bounds;        /* [B][2] max elevation, min elevation */
  int32_t *far_pix;      /* [B*R3D_FAR_CAP] occupied pixels deeper than 500 m, (row << 16) | column */
  int32_t *n_far;        /* [B] */
  int32_t *rebase;       /* [B] re-projections forced so far because the elevation bounds may have moved */
  int32_t *status;       /* [B] R3D_S_* bits */
  /* outputs of r3d_batch_finish -- and SCRATCH between finishes: r3d_batch_begin may use out_xyzi / out_label for the sort of
   * a cloud in no file order (and a diagnostic build for its stamps), a rebase for its passes.  Read the previous batch's
   * results (or order the reads with an event) before the next r3d_batch_begin is enqueued on another stream. */
  float *out_xyzi;       /* [B*cap][4] */
  uint32_t *out_label;   /* [B*cap] */
  int32_t *n_out;        /* [B] */
  void *workspace;
  size_t workspace_bytes;
} r3d_batch_t;

size_t r3d_batch_workspace_bytes(const r3d_batch_t *b);

/* Once per descriptor, before the first r3d_batch_begin: the tables that depend only on the shape
 * (column edges of the projection). */
int r3d_batch_create(const r3d_batch_t *b, void *stream);

/* Step 0 (insertion.py:362, :373-375 for every scene): n_points (device int32[B]) points per
 * scene are already in b->xyzi / b->label.  Computes the elevation bounds, the pixel id of every
 * point and resets all per-scene state.  (The min-reduce of :118-125 is done per insert, on the
 * window of the range image that the insert can see -- DESIGN.md par.3.) */
int r3d_batch_begin(const r3d_batch_t *b, const int32_t *n_points, void *stream);

/* Step 0 from x y z alone (12 bytes per point instead of the 16 of a velodyne row + 4 of a label): xyz3 = device float
 * [B][cap][3], n_points[s] rows used.  For a caller that still holds the frames on the host and takes the batch's DELTA
 * (r3d_batch_export_delta + r3d_host_merge_frames / r3d_host_write_delta_frames): nothing on the device reads a frame
 * point's intensity or label before the compaction, which such a caller does not run.  The slab b->xyzi is filled from
 * xyz3 with intensity 0, b->label is left as it is: r3d_batch_finish and r3d_batch_export_rows' label column are NOT
 * valid after this begin (the inserted points carry their own intensity and label as always). */
int r3d_batch_begin_xyz(const r3d_batch_t *b, const float *xyz3, const int32_t *n_points, void *stream);

/* Step 0 for clouds whose coordinates are genuine float64 (the Waymo flavour: tools/datasets.py:240-262 hands
 * the driver x y z intensity label as float64 after subtracting the LiDAR offset).  rows5: device double
 * [B][cap][5], n_points[s] rows used.  The points are kept like inserted points are -- exact coordinates in the
 * log (rows 0 .. n_points[s]-1, birth step 0), their float32 rounding in xyzi --, so every later step works on
 * the float64 values; r3d_batch_export_rows returns them, and b->log5 rows from n_points[s] on are the accepted
 * visible points (all_visible_parts).  Needs log_cap >= n_points[s] + the points to be inserted
 * (else R3D_S_CAPACITY).  The projection of step 0 uses the reference formula for every point (no float32
 * shortcut: the screen of the fast path assumes float32-exact inputs). */
int r3d_batch_begin_f64(const r3d_batch_t *b, const double *rows5, const int32_t *n_points, void *stream);

/* One placement candidate per scene (insertion.py:453-526).  samples5: rows of [x y z intensity
 * label] float64, scene s owns rows sample_off[s] .. sample_off[s+1] (device int64[B+1]).
 * Scenes with active[s] == 0 (nullable = all active) or an empty sample are skipped.
 * n_visible[s] = len(visible_sample); accepted[s] = 1 iff n_visible >= max(1, min_points[s])
 * (:511-517), in which case the scene is updated exactly as :526 would.  `step` is 1-based and
 * must increase by one per call of an insert slot that may accept.
 *
 * min_points[s] < 0: the state the reference's driver is left with after a REJECTED candidate -- the scene without the
 * points the candidate covers, and without the candidate (insertion.py:468-471 run, :526 not; the copy stays bound to
 * scene_pcl until the next candidate restores the backup, :453, so the next sample's placement search :433 sees it, and
 * when no further candidate comes the next fill_spherical :373 or save_data does).  n_visible[s] as always, accepted[s] =
 * 0; the scene itself is NOT changed: the copy is kept beside it (alive bits only), r3d_batch_export_rows returns it
 * instead of the scene until the scene's next evaluated candidate, and r3d_batch_adopt_rejected makes it the scene.  A
 * driver that wants the reference's behaviour replays a sample's last candidate this way once it has been rejected
 * with n_visible > 0 (INTEGRATION.md par. 6). */
int r3d_batch_insert(const r3d_batch_t *b, const double *samples5, const int64_t *sample_off,
                     const int32_t *min_points, const int32_t *active, int32_t step,
                     int32_t *n_visible, int32_t *accepted, void *stream);

/* The candidate loop of ONE insert slot (insertion.py:449-545) in one call: candidate j (0 <= j < n_cand, as long as
 * first_cand + j < n_possible[s]) of scene s is the rows sample_off[s] .. sample_off[s + 1] of the packed sample list at
 * cand + j * cand_stride (doubles) -- the layout r3d_find_possible_places writes with cand_stride = length of the list.
 * The candidates of a scene are tried in order, the first whose visible part reaches max(1, min_points[s]) is merged exactly as
 * r3d_batch_insert would (all of them as step `step`), accepted_at[s] = first_cand + j; a scene that accepts none keeps its
 * accepted_at[s] (initialise to -1).  n_visible[s] = len(visible_sample) of the last candidate tried.  Scenes with active[s]
 * == 0 (nullable = all active) or an empty sample are skipped.  replay_last != 0: a scene whose LAST existing candidate
 * (first_cand + j + 1 == n_possible[s]) is rejected gets that candidate once more as min_points < 0 -- the state the
 * reference's driver is left with (r3d_batch_insert above, r3d_batch_adopt_rejected).  Same results as one r3d_batch_insert
 * call per candidate with the "still open" scenes as `active`, without the launches and the host's mask arithmetic between. */
int r3d_batch_insert_first(const r3d_batch_t *b, const double *cand, int64_t cand_stride, const int64_t *sample_off,
                           const int32_t *min_points, const int32_t *active, const int32_t *n_possible, int32_t first_cand,
                           int32_t n_cand, int32_t step, int32_t replay_last, int32_t *n_visible, int32_t *accepted_at,
                           void *stream);

/* Materialise the merged clouds: out_xyzi / out_label / n_out = bytes of velodyne/{f}.bin and
 * labels/{f}.label (SS tools/datasets.py:72-84).  check (nullable) float32
 * [B*log_cap][check_cols] = check/{f}.bin rows from the log. */
int r3d_batch_finish(const r3d_batch_t *b, float *check, int32_t check_cols, void *stream);

/* n_slots consecutive insert slots with ONE candidate each (the reference's loop when the first
 * placement of every object is tried, insertion.py:371-545) in one launch in which slot k of a
 * scene starts as soon as slot k-1 of the SAME scene is done, instead of after the slowest scene of
 * the whole batch.  The
 * arguments are HOST arrays of n_slots device pointers with the meaning they have in
 * r3d_batch_insert (active may be null, or hold nulls); slot k runs as step first_step + k.  Same
 * results as n_slots calls of r3d_batch_insert.  The hand-off between the slots of a scene is an
 * agent-scope release / acquire and does not depend on where the workgroups run; the waits are
 * bounded (R3D_S_CHAIN_TIMEOUT: the scene's later slots were not run, its results are not valid);
 * R3D_B_SLOT_LAUNCHES in `reserved` selects one launch per slot unconditionally. */
int r3d_batch_insert_many(const r3d_batch_t *b, int32_t n_slots, const double *const *samples5,
                          const int64_t *const *sample_off, const int32_t *const *min_points,
                          const int32_t *const *active, int32_t first_step, int32_t *const *n_visible,
                          int32_t *const *accepted, void *stream);

/* The current merged cloud of every scene (scene_pcl as find_possible_places gets it, insertion.py:433-434,
 * without the scratch columns): rows4 double [B][cap][4] = x y z label of the living points -- the
 * surviving original points in their order, then the surviving inserted points with their float64
 * coordinates -- and n_rows int32 [B].  Does not change the batch; may be called between inserts. */
int r3d_batch_export_rows(const r3d_batch_t *b, double *rows4, int32_t *n_rows, void *stream);

/* The copy a rejected candidate has left (min_points < 0 above) becomes the scene, for the scenes that hold one and
 * have active[s] != 0 (nullable = all): its bounds and pixels are computed afresh as the next pass of the reference's
 * while-loop would (insertion.py:373-375); b->rebase[s] counts it.  Scenes without such a copy are left alone. */
int r3d_batch_adopt_rejected(const r3d_batch_t *b, const int32_t *active, void *stream);

/* One streaming kernel of the batched path on its own, all scenes, for timing it in isolation
 * with HIP events (bench.py) and for rocprofv3: the state must be the one r3d_batch_begin
 * (BOUNDS, RESET, PROJECT) or r3d_batch_finish (ALIVE_COUNT, ALIVE_WRITE) leaves; every one of
 * them is idempotent on that state. */
#define R3D_K_BOUNDS 1        /* k_bounds: min / max of z/r per scene (insertion.py:74-79) */
#define R3D_K_PREPARE 2       /* k_prepare: bounds from the extremes, row-edge table, living points per tile */
#define R3D_K_PROJECT 3       /* k_project: pixel id of every point (insertion.py:74-76, :104-116) */
#define R3D_K_ALIVE_WRITE 5   /* k_alive_write: survivors, original order, into out_xyzi / out_label */
int r3d_batch_launch_one(const r3d_batch_t *b, int32_t which, void *stream);

/* Diagnostic: the 16 counters the insert kernels keep in the workspace since r3d_batch_create (or the last
 * call with reset != 0), copied to HOST memory; synchronises the stream.  [0] allocations the launch's pool could not
 * serve (those evaluations took a slower route); [1] evaluations whose depth tile lived in the pool (window too large
 * for the workgroup's LDS); [2] pairs evaluated more than once (a predecessor changed a pixel they had read); [3] / [4]
 * evaluations done again for comparison / that differed (descriptor bit 64 of `reserved`: diagnostic, must stay 0);
 * [5] wave-rounds of the gather whose hits did not fit the workgroup's LDS (their coordinates were fetched right away);
 * [6] scenes handed to k_insert_big; [7] rebases inside the chain kernel; [8..11] rebases by reason: a visible sample point
 * outside the elevation bounds / a culled point held a bound / the same found by the far-pixel pass / none of these (must stay 0);
 * [12..15] index checks a diagnostic build (-DR3D_CHECK) saw fail.  reset: bit 0 clears the counters after the copy; bit 1:
 * host_out16 holds 32 values, [16..31] are the notes such a build keeps about the first failed check; bit 2: it holds 64
 * values, [32..] are round 5's counters: [32] pairs committed by the workgroup that evaluated them, [33] chunks those pairs
 * listed in all, [34] / [35] pairs whose evaluator found their predecessors still at work and left them -- with a record of
 * the evaluation / as they came -- to the workgroup that finishes slot k - 1 (nobody waits: csrc/r3d_insert.hip), [36] pairs
 * committed from such a record, [37] scenes whose points were put into virtual order at step 0 (below); round 6: [38] evaluations on a
 * sparse depth tile (a window beyond the LDS keeps only the pixels the evaluation reads), [39] ... that it could not hold either
 * (pool), [40] / [41] scene / sample points whose pixel the reference formula decided with the fractional row or column position
 * (insertion.py:104-105 before int()) within 1e-12 of an integer -- where the device library's arctan2 / arccos and NumPy's, an
 * ULP apart, could truncate to neighbouring bins; every point the verified fast projection cannot confirm is looked at;
 * [42] scenes begun under R3D_B_FILE_ORDER whose chunk boxes say that their points come in no file order (the promise then
 * costs time: every insert walks the whole cloud). */
int r3d_batch_debug_counters(const r3d_batch_t *b, int32_t *host_out16, int32_t reset, void *stream);

/* Point order.  The incremental state of Level 2 is kept per 64 consecutive points (alive word, bounding box of their
 * pixels), which is cheap when consecutive points are neighbours in the range image -- every LiDAR file order is like that:
 * ring-major (KITTI, SemanticKITTI), firing sequence by firing sequence (all lasers of one azimuth, then the next).  The
 * reference does not care about the order (insertion.py:100-127 loops over the points as they come), and neither do the
 * results here: r3d_batch_begin looks at the boxes it has just built and, for a scene whose mean box exceeds 1 024 pixels (and
 * that has 4 096 points or more), numbers the points anew by (row, band of 64 columns) -- internally: slabs, log and the
 * order of every output are untouched.  Bit 1024 of `reserved`: every scene of this batch (tests).  The look costs a begin four small launches (~20 us per 256 scenes); a caller
 * who knows that its clouds come in a file order says so with R3D_B_FILE_ORDER in `reserved` and saves them -- a cloud that
 * does not keep the promise costs time (every insert then walks the whole cloud), never results.  The bit belongs to the
 * batch from one r3d_batch_begin to the next: finish / export_delta / export_rows skip the launch that puts the alive bits of
 * re-numbered scenes back into slab order when it is set, so it must not be set between a begin that looked and its finish
 * (a scene that was numbered anew comes back flagged R3D_S_ORDER_PROMISE if it is).
 * The Python mirror (SceneBatch) sets the bit by itself, at begin, once a batch has come through without an unordered scene,
 * and looks again now and then.
 *
 * r3d_batch_export_pix: the pixel id of every point of every scene, in the order of the slabs, as the reference numbers
 * it (row * cols + column, insertion.py:116; `pix` itself holds (row << 16) | column in the internal numbering). */
int r3d_batch_export_pix(const r3d_batch_t *b, int32_t *pix_ids /* [B*cap] */, void *stream);
/* r3d_batch_export_alive: the alive word of every 64-point chunk of every scene in the order of the slabs, alive [B][chunks]
 * (chunks = (cap + 63) / 64; bits beyond a scene's n_total are 0) -- the living points of the cloud r3d_batch_export_rows
 * shows (the copy a rejected candidate has left while there is one), without the rows: what R3D_PQ_SCENE_SLAB queries of the
 * placement search read.  Does not change the batch. */
int r3d_batch_export_alive(const r3d_batch_t *b, uint64_t *alive, void *stream);
/* r3d_batch_point_order: per scene, how many of its points the last r3d_batch_begin numbered anew (0: the scene keeps the order
 * of its slab), into DEVICE memory -- what a caller that sets R3D_B_FILE_ORDER by itself looks at (SceneBatch does). */
int r3d_batch_point_order(const r3d_batch_t *b, int32_t *virtual_order /* [B], device */, void *stream);

/* =====================================================================================
 * Level 3 -- placement search.
 *
 * Replaces find_possible_places of semantic_segmentation/Real3DAug/tools/find_spot.py:192-273 with
 * its helpers rotate_bounding_box_2 (:42-76), check_bounding_box (:79-104), correct_height
 * (:107-152) and cut_bounding_box (tools/cut_bbox.py:7-68): the sample and its box are turned
 * around the sensor in 360 steps of one degree; a step is a possible placement when every sample
 * point that falls inside the rich map lies on an allowed cell (:234-248), placement surface is
 * found within the growing search radius (the box is then put on its mean height, and stays
 * there for the following steps), no scene point other than placement surface is inside the
 * box and no sample point is inside an annotated scene box.
 *
 * One query = one sample tried in one scene; queries are independent and run together.  The
 * descriptors live in DEVICE memory, like every array they point to.
 * ===================================================================================== */
#define R3D_PLACE_ROTATIONS 360       /* find_spot.py:228 */
#define R3D_PLACE_MAX_OK_LABELS 32      /* (round 6: 8 before -- a config may list more placement labels per class than the reference's) */
#define R3D_PLACE_SURFACE_CAP 128     /* surface points per step whose heights can be summed in order when their sum
                                         depends on the order (never for float32 LiDAR heights of similar size) */
#define R3D_PLACE_MAX_RADII 64

#define R3D_PS_SURFACE_OVERFLOW 1     /* more than R3D_PLACE_SURFACE_CAP surface points in the search radius AND heights
                                         whose sum depends on the order of addition */
#define R3D_PS_NONFINITE 2            /* NaN / Inf in the sample, its box or the pose */
#define R3D_PS_BAD_DESCRIPTOR 4       /* the query's descriptor cannot be followed: a null or misaligned pointer, an address no
                                         device allocation can have, a negative count, m outside 1..8192, label columns outside
                                         the row, ...  Looked at on the device before any kernel follows a pointer: the query
                                         gets no placements (n_possible 0), the other queries of the call are served */

/* flags[q][r-1] bits */
#define R3D_PF_ON_SURFACE 1           /* :234-248 passed */
#define R3D_PF_NEAR_ROAD 2            /* correct_height found surface (:251-254) */
#define R3D_PF_SCENE_IN_BOX 4         /* a non-surface scene point inside the sample's box (:91-97) */
#define R3D_PF_SAMPLE_IN_BOX 8        /* a sample point inside an annotated scene box (:99-103) */
#define R3D_PF_POSSIBLE 16

/* r3d_place_query_t.flavour: the object_detection tree's find_spot.py differs from the semantic_segmentation one in */
#define R3D_PQ_POINTWISE_ROTATION 1   /* points turned one by one (OD find_spot.py:94-99: np.dot per point, another BLAS
                                         accumulation order than the SS matrix product, :72) */
#define R3D_PQ_MAP_NEEDS_POINT 2      /* OD :261-275: map position = x, y - map_move (identity pose rows), inside the map
                                         when 0 <= position < size, and at least one sample point must be inside */
#define R3D_PQ_COLLIDE_LABEL 4        /* OD :120-121: scene points of label collide_label collide (SS :92-97: every label
                                         that is not placement surface) */
#define R3D_PQ_COLLIDE_ABOVE 8        /* OD :123-124 ('Pedestrian'): only points with z >= box bottom + collide_dz */
#define R3D_PQ_SCENE_SLAB 16          /* the current cloud is given as a scene of a batch (r3d_batch_t) stands in HBM instead of as
                                         float64 rows: `scene` = its float32 x y z intensity rows (b.xyzi + s*cap*4, cast), n_scene =
                                         n_total[s], scene_head = n_head[s], scene_label = b.label + s*cap, scene_alive = the scene's
                                         alive words (r3d_batch_export_alive), scene_tail_ref / scene_log5 = b.tail_ref / b.log5 of the
                                         scene (the float64 coordinates of inserted points); dead points are skipped.  Saves the
                                         export of the rows (r3d_batch_export_rows) and their chunk ranges per insert slot */
#define R3D_PQ_ORIG_SLAB 32           /* the original cloud likewise: `orig` = float32 x y z intensity rows (16 bytes each, e.g. the
                                         first n_head[s] rows of a batch's scene: they are never moved or changed), orig_label =
                                         their label words (& 0xFFFF taken here); orig_ld / orig_label_col unused; orig_ranges from
                                         r3d_places_chunk_ranges_f32.  Same values as the float64 rows [x y z label] of the same
                                         points, half the bytes, no copy */

typedef struct r3d_place_query_t {
  const double *scene;     /* current cloud, n_scene rows of scene_ld doubles: x y z at columns 0-2, the label at
                              scene_label_col (scene_pcl N x 9, label column 7, insertion.py:433) */
  const double *orig;      /* original cloud (height search, find_spot.py:123-131): original_pcl N x 5, label column 4 */
  const double *boxes;     /* [n_boxes][10] annotated scene objects: centre x y z, quaternion x y z w, length, width, height */
  const double *sample;    /* [m][5] x y z intensity label (sample_data['pcl']) */
  const uint8_t *map;      /* [map_rows][map_cols] rich map */
  const float *scene_ranges, *orig_ranges; /* nullable: r3d_places_chunk_ranges() of the two clouds; lets the search
                              skip the 64-point chunks that are out of reach (same results either way) */
  int64_t n_scene, n_orig;
  int32_t scene_ld, scene_label_col, orig_ld, orig_label_col;
  int32_t n_boxes, m, map_rows, map_cols;
  int32_t n_ok_labels;     /* labels an object of this class may stand on, in config order (:218-223) */
  int32_t cand_cap;        /* room for this many candidate clouds at cand + cand_off */
  int32_t ok_labels[R3D_PLACE_MAX_OK_LABELS];
  uint64_t ok_map[4];      /* bit v set: map value v is an allowed surface */
  double anno[10];         /* the sample's box after read_label_line (:155-189): centre, quaternion, length, width, height */
  double pose[8];          /* rows 0 and 1 of the 4x4 pose (transformation_matrix) */
  double map_move[2];      /* map_move[0], map_move[1] */
  int64_t cand_off;        /* in doubles */
  int64_t cand_stride;     /* doubles between consecutive candidate clouds of this query (>= m*5): with
                              cand_off = start of the query's rows in a packed list of all queries and
                              cand_stride = length of that list, candidate j of ALL queries is one packed
                              sample list at cand + j*cand_stride -- what r3d_batch_insert takes */
  int32_t flavour;         /* R3D_PQ_* bits, 0 = semantic_segmentation */
  int32_t collide_label;
  double collide_dz;
  /* R3D_PQ_SCENE_SLAB only (else unused): */
  const uint32_t *scene_label;    /* [n_scene] label words of the scene's points (& 0xFFFF taken here) */
  const uint64_t *scene_alive;    /* [(n_scene + 63) / 64] bit i of word c: point 64 c + i lives */
  const int32_t *scene_tail_ref;  /* [n_scene - scene_head] log row of point scene_head + t */
  const double *scene_log5;       /* rows of 5 doubles: x y z of the inserted points */
  int64_t scene_head;             /* float32-exact points at the front of the scene */
  /* R3D_PQ_ORIG_SLAB only (else unused): */
  const uint32_t *orig_label;     /* [n_orig] label words of the original cloud's points */
} r3d_place_query_t;

size_t r3d_places_workspace_bytes(int32_t n_queries, int32_t max_boxes);

/* ranges float [ceil(n/64)][2]: smallest and largest distance from the sensor's z axis among the points of
 * every 64-point chunk of a cloud (rows of ld doubles, x y first). */
int r3d_places_chunk_ranges(const double *rows, int64_t n, int32_t ld, float *ranges, void *stream);
/* The same for float32 rows [x y z intensity] of 16 bytes (R3D_PQ_SCENE_SLAB / R3D_PQ_ORIG_SLAB): the ranges of the same
 * points given as float64 rows, bit for bit. */
int r3d_places_chunk_ranges_f32(const float *rows4, int64_t n, float *ranges, void *stream);

/* radius_sq (HOST array): radius**2 of the search steps that can still succeed (find_spot.py:121-140:
 * 0.1, 0.1+0.1, ... while the next radius is <= 5), as the caller's interpreter evaluates them.
 * Outputs (device): flags uint8 [Q][360]; n_possible int32 [Q]; rot_out int32 [Q][360] = rotation
 * numbers (1..360) of the possible placements in order; anno_out double [Q][360][7] = box centre
 * and quaternion of every possible placement, same order; cand = for placement ordinal j in
 * [first_cand, first_cand + cand_cap) the m x 5 cloud at cand + cand_off + (j - first_cand)*cand_stride
 * (deepcopy(sample_pcl), :258-264); status int32 [Q] (R3D_PS_*). */
/* The one thing the library keeps between calls besides the thread-local error string (and caches of device attributes): a
 * helper stream with two events per (device, caller's stream) on which r3d_find_possible_places runs its orientation chain
 * beside the point passes, made on first use.  r3d_places_release() waits for them, destroys them and forgets them (call it
 * when no r3d_find_possible_places is in flight; the next call makes them again). */
int r3d_places_release(void);

int r3d_find_possible_places(const r3d_place_query_t *queries, int32_t n_queries, int64_t max_n_scene,
                             int64_t max_n_orig, int32_t max_m, int32_t max_boxes, const double *radius_sq,
                             int32_t n_radii, uint8_t *flags, int32_t *n_possible, int32_t *rot_out,
                             double *anno_out, double *cand, int32_t first_cand, int32_t *status,
                             void *workspace, size_t workspace_bytes, void *stream);

/* =====================================================================================
 * cut_bounding_box for many boxes of one cloud.
 *
 * Replaces tools/cut_bbox.py:7-68 (strict = 1: strictly inside the six faces) and the box half of
 * separate_bbox :71-123 (strict = 0: a point is outside when beyond a face, so points on a face
 * stay in the box) as the object-database creation calls them once per annotated object of a
 * frame (cut_object/cut_out.py:100-157: the points inside the box whose label is the object's).
 * rows: n rows of ld doubles, x y z first, the label at label_col (< 0: no label filter);
 * boxes10 [k][10] = centre, quaternion xyzw, length, width, height (annotation_move already
 * subtracted from the centre); box_labels [k] (nullable, NaN = any label).  counts [k] = number
 * of points inside; index [k][index_cap] = their row numbers in cloud order (entries beyond
 * index_cap are dropped, counts says how many there are).
 * ===================================================================================== */
size_t r3d_cut_boxes_workspace_bytes(int64_t n, int32_t k);
int r3d_cut_boxes(const double *rows, int64_t n, int32_t ld, int32_t label_col, const double *boxes10,
                  const double *box_labels, int32_t k, int32_t strict, int32_t *counts, int32_t *index,
                  int64_t index_cap, void *workspace, size_t workspace_bytes, void *stream);

/* =====================================================================================
 * Rich-map rasterisation: the __main__ block of
 * semantic_segmentation/rich_map/drivable_area_map.py:122-206 for one sequence.
 *
 * xyzi / label: a frame as read from its velodyne .bin and labels .label files; pose16: the frame's
 * row-major 4x4 transform_matrix (tools/datasets.py:62-67).
 * r3d_map_bounds: folds the frame's world x / y extremes into minmax[4] = {min x, max x, min y,
 *   max y} as order-preserving keys (:143-150); initialise to {~0, 0, ~0, 0}, decode with
 *   key >> 63 ? key & ~(1 << 63) : ~key  (r3d_device.hpp ordered_key).
 * r3d_map_splat: :172-200 for one frame into keys[size_x * size_y] (zero-initialised, shared by
 *   the frames of the sequence); frame_no = the frame's position in processing order.  Labels
 *   are config['insertion']['placement_labels'][1], [2], [3] (road, sidewalk, parking).
 *   status bit 0: a surface point outside the map (the reference's assert :180).
 * r3d_map_finish: keys -> map values 0..3 as float64 (what np.savez stores, :205) and / or uint8.
 * ===================================================================================== */
int r3d_map_bounds(const float *xyzi, int64_t n, const double *pose16, uint64_t *minmax, void *stream);
int r3d_map_splat(const float *xyzi, const uint32_t *label, int64_t n, const double *pose16,
                  const int32_t *labels_road, int32_t n_road, const int32_t *labels_sidewalk, int32_t n_sidewalk,
                  const int32_t *labels_parking, int32_t n_parking, double min_x, double min_y, int32_t size_x,
                  int32_t size_y, int64_t frame_no, uint64_t *keys, int32_t *status, void *stream);
int r3d_map_finish(const uint64_t *keys, int64_t cells, double *map64, uint8_t *map8, void *stream);

/* The object-detection flavour, object_detection/rich_map/single_drivable_area_map.py:113-194, for ONE frame
 * (KITTI has no poses: a map per frame, in the frame's own coordinates).  min_x / min_y = int(min x), int(min y)
 * of the frame's points and size_x / size_y = int(max) + 1 - min (:118-127; r3d_map_bounds with the identity
 * pose gives the extremes).  road_map [size_x][size_y] uint8 = cells under points of label road_label (:129-139),
 * closed with disk(4) (:145-151; {0, 1} as np.savez stores it, :157); pedestrian_map = the cells that are not road
 * but touch it (:160-178), dilated with disk(2) (:180-188, saved :193).  scratch: 2*size_x*size_y bytes. */
int r3d_od_maps(const float *xyzi, const uint32_t *label, int64_t n, int32_t road_label, int32_t min_x, int32_t min_y,
                int32_t size_x, int32_t size_y, uint8_t *road_map, uint8_t *pedestrian_map, uint8_t *scratch,
                void *stream);

/* =====================================================================================
 * Host-side packer of the file-to-file driver.  HOST pointers: copies B
 * frames as the reference's __getitem__ reads them (velodyne .bin rows float32 x y z intensity,
 * .label words; SS tools/datasets.py:51-56) into the slabs of a (pinned) staging buffer laid out
 * like r3d_batch_t.xyzi / .label ([B][cap]); labels are masked with 0xFFFF (:53-55), or, with
 * collapse_keep >= 0, collapsed to {collapse_keep, 1} (OD insertion.py:353-355).  `threads` host
 * threads share the frames.  The copy into HBM is the caller's.
 * ===================================================================================== */
int r3d_host_pack_frames(const float *const *xyzi, const uint32_t *const *label, const int32_t *n_points, int32_t B,
                         int64_t cap, float *dst_xyzi, uint32_t *dst_label, int32_t collapse_keep, int32_t threads);
/* The same, and x y z of every point once more, 12 bytes per point, into dst_xyz3 [B][cap][3] (NULL: not wanted): what a
 * caller in delta mode uploads (r3d_batch_begin_xyz) -- the intensities and labels of the frames stay on the host, where the
 * files are written from the staging slab (r3d_host_write_delta_frames, r3d_host_merge_frames). */
int r3d_host_pack_frames_xyz(const float *const *xyzi, const uint32_t *const *label, const int32_t *n_points, int32_t B,
                             int64_t cap, float *dst_xyzi, uint32_t *dst_label, float *dst_xyz3, int32_t collapse_keep,
                             int32_t threads);

/* The delta of a batch instead of its merged clouds, for a caller that still holds the frames on the host (the
 * streamed file-to-file driver): alive [B][chunks] uint64 -- bit i of word c = point 64 c + i of the scene survives
 * (chunks = (cap + 63) / 64; bits beyond the scene's count are 0) --, the inserted points in insertion order,
 * tail_xyzi [B][tail_stride][4] float32 (what save_data writes of them, SS tools/datasets.py:81) and tail_label
 * [B][tail_stride], and counts [2][B] int32 = points of the frame, points after the inserts.  Called after the
 * inserts INSTEAD of r3d_batch_finish (no compaction runs on the device); does not change the batch.
 * Not for batches begun with r3d_batch_begin_f64. */
int r3d_batch_export_delta(const r3d_batch_t *b, uint64_t *alive, float *tail_xyzi, uint32_t *tail_label, int64_t tail_stride,
                           int32_t *counts, void *stream);

/* HOST pointers: the merged clouds from the frames as packed by r3d_host_pack_frames (in_xyzi / in_label, [B][cap])
 * and the delta above (copied to the host): out_xyzi [B][out_cap][4], out_label [B][out_cap], n_out [B] -- the bytes
 * of velodyne/{f}.bin and labels/{f}.label (insertion.py:472-473, :526; SS tools/datasets.py:80-84) -- and, when
 * check != NULL, check [B][check_stride][check_cols] = every inserted point, x y z intensity (label) as float32
 * (the bytes of check/{f}.bin, :73-75, :86-88).  `threads` host threads share the frames. */
int r3d_host_merge_frames(const float *in_xyzi, const uint32_t *in_label, int64_t cap, const uint64_t *alive, int64_t chunks,
                          const float *tail_xyzi, const uint32_t *tail_label, int64_t tail_stride, const int32_t *counts,
                          int32_t B, float *out_xyzi, uint32_t *out_label, int64_t out_cap, int32_t *n_out, float *check,
                          int64_t check_stride, int32_t check_cols, int32_t threads);

/* HOST: the files of B frames straight into / out of the staging slabs, by `threads` native threads (a Python reader per
 * file serialises on the interpreter lock).  r3d_host_read_frames = what SemanticKITTI.__getitem__ / KITTI.__getitem__ read
 * (SS tools/datasets.py:51-56: velodyne/{f}.bin float32 rows of 4, labels/{f}.label uint32) into the layout of
 * r3d_host_pack_frames (labels masked with 0xFFFF, or collapsed with collapse_keep >= 0; label_paths NULL: labels 0), with
 * n_points [B] = rows of every file.  r3d_host_write_frames = the files save_data stores (SS :80-89, OD :86-93): float32
 * rows of 4, uint32 labels (label_paths NULL or a NULL entry: none), check rows; each under "<path>.tmp" first, then
 * renamed, check last; a NULL velodyne path skips the frame.  Errors name the file in r3d_last_error(): R3D_E_ARG a file that
 * is not what a reader expects / counts beyond the buffers, R3D_E_IO a write that failed. */
int r3d_host_read_frames(const char *const *velodyne_paths, const char *const *label_paths, int32_t B, int64_t cap, float *dst_xyzi,
                         uint32_t *dst_label, int32_t *n_points, int32_t collapse_keep, int32_t threads);
int r3d_host_read_frames_xyz(const char *const *velodyne_paths, const char *const *label_paths, int32_t B, int64_t cap, float *dst_xyzi,
                             uint32_t *dst_label, float *dst_xyz3 /* [B][cap][3], nullable: see r3d_host_pack_frames_xyz */,
                             int32_t *n_points, int32_t collapse_keep, int32_t threads);
int r3d_host_write_frames(const char *const *velodyne_paths, const char *const *label_paths, const char *const *check_paths, int32_t B,
                          const float *xyzi, const uint32_t *label, int64_t cap, const int32_t *n_out, const float *check,
                          int64_t check_stride, int32_t check_cols, const int32_t *n_check, int32_t threads);

/* HOST: the same files straight from what the host already holds -- the frames in the staging slab the upload was made from
 * (in_xyzi, in_label: the layout of r3d_host_pack_frames / r3d_host_read_frames) and the delta of r3d_batch_export_delta
 * (alive, tail_xyzi, tail_label, counts) -- without the merged clouds in between: velodyne and label files are written
 * run by run of surviving points (writev), then the surviving inserted points; check/{f}.bin = every inserted point.  Byte
 * for byte what r3d_host_merge_frames followed by r3d_host_write_frames writes; n_out [B] (may be NULL) = points per
 * merged cloud.  label_paths / check_paths / their entries may be NULL as there. */
int r3d_host_write_delta_frames(const char *const *velodyne_paths, const char *const *label_paths, const char *const *check_paths,
                                int32_t B, const float *in_xyzi, const uint32_t *in_label, int64_t cap, const uint64_t *alive,
                                int64_t chunks, const float *tail_xyzi, const uint32_t *tail_label, int64_t tail_stride,
                                const int32_t *counts, int32_t check_cols, int32_t *n_out, int32_t threads);

/* HOST: object_detection/Real3DAug/tools/datasets.py:20-37 (create_annotation, called by save_data :81-84) for the n frames
 * of a batch: dst[i] = the bytes of the frame's label_2 file src[i] followed by extra[i] (the lines of the inserted objects,
 * one zero-terminated string per frame; NULL adds nothing), written to dst[i].tmp and renamed.  A NULL dst[i] is skipped.
 * The reference opens both files in text mode: "\r\n" and a lone "\r" of the source arrive as "\n" (universal newlines) and
 * are written as "\n" -- so they are here. */
int r3d_host_append_text_files(const char *const *src, const char *const *dst, const char *const *extra, int32_t n,
                               int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* REAL3DAUG_HIP_H */
