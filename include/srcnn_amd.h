/*
 * srcnn_amd.h -- C ABI of the MI355X-native SRCNN Y-channel path (libsrcnn_amd.so).
 *
 * This is the drop-in boundary for the ONE hot path of rageworx/libsrcnn: the 2x "bicubic"
 * (Mitchell) upscale of the Y plane followed by the three fixed-weight convolutions
 * (9x9x1->64 +ReLU, 1x1x64->32 +ReLU, 5x5x32->1 +clamp).  Plain pointers and sizes only; no
 * torch / C++ types.  Every entry point cites the reference interface it stands in for
 * (paths relative to the reference tree, rageworx/libsrcnn v0.1.10.40).
 *
 * The reference's public API is C++-linkage (src/libsrcnn.h:46-54: reference parameters and a
 * default argument).  The same shared object therefore ALSO exports the two C++ symbols
 *     void ConfigureFilterSRCNN(SRCNNFilterType, bool)
 *     int  ProcessSRCNN(const unsigned char*, unsigned, unsigned, unsigned, float,
 *                       unsigned char*&, unsigned&, unsigned char**, unsigned*)
 * declared in include/libsrcnn_dropin.h, with identical mangled names, argument meaning,
 * ownership (caller delete[]s) and return codes -- see INTEGRATION.md.
 *
 * All functions return 0 on success or a negative SRCNN_E_* code; srcnn_last_error() gives the
 * text for the calling thread.  There is NO CPU fallback: without a gfx950 device every compute
 * entry point fails with SRCNN_E_NODEVICE.
 *
 * Numerics: mode SRCNN_MODE_STRICT (default) reproduces the reference's float32 results bit
 * for bit (same operation order, separate multiply and add roundings, fp64 where the reference
 * uses double).  The SRCNN_MODE_FAST* tiers contract multiply-add pairs (max |dY| ~2e-4 on the
 * 0..255 scale vs the reference -- the size of the reference's own fp32 rounding noise) and are never
 * used unless asked for.
 */
#ifndef SRCNN_AMD_H
#define SRCNN_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define SRCNN_AMD_ABI_VERSION 5   /* FROZEN at 5 since round 6.  This header is the STABLE ABI: the functions it declares are
                                   * listed in include/srcnn_amd.abi (one name per line), tests/test_abi.py fails when the two or the
                                   * library's export table disagree, so an addition is a deliberate edit of that list and a bump here.
                                   * Instruments (test hooks, diagnostics, the relaxation experiment) live in srcnn_amd_debug.h
                                   * and carry no compatibility promise.
                                   * 2: srcnn_comm_gatherv_f32, srcnn_comm_rank
                                   * 3: contexts (srcnn_init_devices ...), node-level calls, srcnn_trim, sub-band gather
                                   * 4: srcnn_process_u8_begin/_wait, srcnn_comm_wait / srcnn_comm_set_timeout_ms
                                   * 5: gather tables verified across the ranks by default */

/* error codes.  -1/-2/-11/-12/-100 are the reference's own (src/libsrcnn.cpp:951-966,883,910,636) */
#define SRCNN_OK            0
#define SRCNN_E_ARG        -1   /* NULL / zero-sized argument                    (libsrcnn.cpp:951-952) */
#define SRCNN_E_SCALE      -2   /* non-positive scaled size                      (libsrcnn.cpp:963-966) */
#define SRCNN_E_OUTALLOC  -11   /* output buffer allocation failed               (libsrcnn.cpp:883)     */
#define SRCNN_E_CONVALLOC -12   /* conv-Y buffer allocation failed               (libsrcnn.cpp:910)     */
#define SRCNN_E_NORESULT -100   /* "no RGB produced" default                     (libsrcnn.cpp:636,968) */
#define SRCNN_E_NODEVICE -200   /* no gfx950 device / HIP runtime unusable */
#define SRCNN_E_HIP      -201   /* a HIP call failed; see srcnn_last_error() */
#define SRCNN_E_DEVMEM   -202   /* device allocation failed */
#define SRCNN_E_UNSUPPORTED -203
#define SRCNN_E_COMM     -204   /* RCCL failure */

/* filter ids == SRCNNFilterType (src/libsrcnn.h:37-44) */
#define SRCNN_FILTER_NEAREST  0
#define SRCNN_FILTER_BILINEAR 1
#define SRCNN_FILTER_BICUBIC  2
#define SRCNN_FILTER_LANCZOS3 3
#define SRCNN_FILTER_BSPLINE  4

#define SRCNN_MODE_STRICT   0   /* default: bit-identical to the reference */
#define SRCNN_MODE_FAST     1   /* fp32 FMA chains (layers 1+2 on the fp32 MFMA, layer 3 v_fma_f32) */
#define SRCNN_MODE_FAST_F16 2   /* layers 1+2 as split-fp16 GEMMs on the fp16 matrix pipe (3 MFMAs per product
                                 * term set, fp32 accumulate); same error class as SRCNN_MODE_FAST */
/* ---- lifecycle (the reference has none: it is stateless CPU code; src/libsrcnn.cpp:91-92 are its
 *      only globals).  srcnn_init is idempotent and thread-safe; every compute call self-inits on
 *      device 0 if it was never called.
 *
 * Threading contract.  Every entry point may be called from any host thread at any time.
 *   - srcnn_process_u8 / ProcessSRCNN are re-entrant like the reference's (src/libsrcnn.cpp:628-923 allocates
 *     everything per call): a call leases a private lane (streams, scratch, staging) for its duration; up to 4 per
 *     context (env SRCNN_MAX_LANES) run concurrently, further callers wait for a lane.
 *   - *_dev calls take their scratch from the given stream's workspace; two threads using the SAME stream are
 *     serialised while they enqueue, different streams are independent.
 *   - The numerics mode is sampled once when a call starts (for the asynchronous pair: in srcnn_process_u8_begin); srcnn_set_mode
 *     never affects a call in flight.  A strict-only build of the library (make STRICT_ONLY=1) refuses every mode but
 *     SRCNN_MODE_STRICT with SRCNN_E_UNSUPPORTED.
 *   - srcnn_stream_destroy / srcnn_batch_graph_destroy / srcnn_shutdown must not race with calls that still use
 *     that stream / graph / the library (as with any handle).
 *
 * Contexts (one process, several GPUs).  A context is one device binding with everything that lives in that device's
 * memory (weights, table cache, workspaces, lanes).  srcnn_init(device) creates context 0 -- the one-device-per-process
 * model of `torchrun`-style launches.  srcnn_init_devices(list, n) creates n contexts, list[k] = HIP device of context k
 * (NULL / n <= 0: one context per visible device; a device may be listed more than once -- "virtual contexts", used to
 * test the node-level paths on a one-GPU machine).  A process that never calls either self-initialises from the
 * environment: SRCNN_DEVICES=all | <comma list of device ids>, default device 0.
 *   - With more than one context, the calls that take HOST memory use the whole node on their own: srcnn_process_u8 /
 *     ProcessSRCNN deal the bands of a large image to the contexts (the reference's one call saturates its machine through
 *     OpenMP, src/libsrcnn.cpp:665,791-798,817-824), srcnn_y_upscale2x_f32_stream deals the frames.
 *   - Calls that take DEVICE pointers run on the context that owns the given stream (srcnn_stream_create remembers it),
 *     or, for the NULL stream, on the calling thread's current context: srcnn_set_context(k), default 0 (thread-local,
 *     like hipSetDevice).  Allocation / event / sync plumbing also follows the current context. ---- */
int         srcnn_abi_version(void);
int         srcnn_device_count(void);              /* number of visible HIP devices (0 if none) */
int         srcnn_init(int device);                /* bind context 0 to `device` (< 0: SRCNN_DEVICES or device 0), upload weights */
int         srcnn_init_devices(const int* devices, int n);   /* one context per entry; NULL / n <= 0: every visible device */
int         srcnn_context_count(void);
int         srcnn_context_device(int k);           /* HIP device of context k, or < 0 */
int         srcnn_set_context(int k);              /* current context of the calling thread; returns the previous one or < 0 */
int         srcnn_get_context(void);
void        srcnn_shutdown(void);                  /* free workspaces, streams, comm, all contexts */
int         srcnn_trim(void);                      /* give back what idle lanes and unreferenced cache entries hold */
const char* srcnn_last_error(void);
int         srcnn_set_mode(int mode);              /* SRCNN_MODE_*; returns previous mode or <0 */
int         srcnn_get_mode(void);
int         srcnn_device_name(char* buf, size_t cap);
/* Upper bound, in bytes, on the layer-2 scratch (128 B per output pixel) one pass may hold; larger frames / bands
 * are produced in horizontal sub-bands with identical results.  Default 4.5 GiB or env SRCNN_MAX_WORKSPACE_MB: a
 * 3840x2160 -> 7680x4320 frame (4.25 GB of layer-2 planes) still runs as one pass.  Where memory is short the cap can go
 * down a long way: 2 GiB (two bands) costs -1...+2 % depending on the box, 512 MiB +2 %, 256 MiB +9 %; below that the short
 * bands fill the persistent grid badly (128 MiB +36 %; profiles/r06_lowmem.txt).
 * Returns the previous limit.  Applies to calls that start afterwards, including the bands of srcnn_process_u8.  A band is
 * never smaller than 16 rows (one tile row of the layer kernels), so a limit below 16 rows' worth is exceeded, not refused. */
size_t      srcnn_set_workspace_limit(size_t bytes);

/* ---- device memory / stream / event plumbing so callers need no HIP headers ---- */
void* srcnn_dev_alloc(size_t bytes);               /* NULL on failure */
void  srcnn_dev_free(void* p);
void* srcnn_host_alloc_pinned(size_t bytes);
void  srcnn_host_free_pinned(void* p);
int   srcnn_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);  /* async if stream!=NULL && pinned */
int   srcnn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int   srcnn_memset_dev(void* dst, int byte, size_t bytes, void* stream);
int   srcnn_stream_create(void** stream);
int   srcnn_stream_destroy(void* stream);
int   srcnn_stream_sync(void* stream);             /* NULL = default stream */
int   srcnn_device_sync(void);
int   srcnn_event_create(void** ev);
int   srcnn_event_destroy(void* ev);
int   srcnn_event_record(void* ev, void* stream);
int   srcnn_stream_wait_event(void* stream, void* ev);             /* later work on `stream` waits for `ev` (device side) */
int   srcnn_event_elapsed_ms(void* start, void* stop, float* ms);   /* syncs on `stop` */

/* ---- THE HOT PATH, device-resident -------------------------------------------------------
 * Replaces, for plane 0 (Y), the sequence in libsrcnn::doSRCNN:
 *   FRAWResizeEngine::scale with FRAWBicubicFilter      src/libsrcnn.cpp:716-723, src/frawscale.cpp:162-385
 *   64 x convolution99                                   src/libsrcnn.cpp:785-798 (:350-422)
 *   32 x convolution11                                   src/libsrcnn.cpp:811-824 (:424-447)
 *   convolution55                                        src/libsrcnn.cpp:838-846 (:449-529)
 * d_in : planar float32 Y, w*h, row-major, values nominally 0..255 (device memory)
 * d_out: planar float32 Y', (2w)*(2h) (device memory)
 * Launches asynchronously on `stream` (NULL = default stream); scratch comes from a grow-only
 * per-stream workspace owned by the library, so steady-state calls do no allocation.  The scratch is
 * 128 B per output pixel; above a budget (env SRCNN_MAX_WORKSPACE_MB, default 4608) the frame is
 * produced in horizontal bands internally, with identical results.  Limits: 2^20 output rows, 2^31 pixels. */
int srcnn_y_upscale2x_f32_dev(const float* d_in, unsigned w, unsigned h, float* d_out, void* stream);

/* Same for `nframes` frames stored back to back (config "batch of 64 1080p frames" /
 * "stream of 4K frames").  Frames are independent: identical to nframes single calls. */
int srcnn_y_upscale2x_f32_batch_dev(const float* d_in, unsigned w, unsigned h, unsigned nframes,
                                    float* d_out, void* stream);

/* The batch call captured once into a hipGraph and replayed (config "stream of frames ... with per-GPU hipGraph
 * capture"): create() runs the batch eagerly once (tables, workspaces), captures the same launches for these
 * exact buffers on `stream`, and returns a handle; launch() replays it on that stream.  Results are those of
 * srcnn_y_upscale2x_f32_batch_dev.  (On MI355X the kernels are milliseconds long, so replay and eager launches
 * measure the same; the entry point exists for callers whose frames are small.) */
/* Lifetime: the handle owns a private scratch workspace and references to the contribution tables its kernel nodes
 * use, so it stays valid whatever else runs on `stream` afterwards (larger frames, other graphs); it only needs
 * `stream`, d_in and d_out to outlive it.  Destroy it before srcnn_stream_destroy(stream) / srcnn_shutdown. */
int srcnn_batch_graph_create(const float* d_in, unsigned w, unsigned h, unsigned nframes, float* d_out, void* stream,
                             void** graph);
int srcnn_batch_graph_launch(void* graph);
int srcnn_batch_graph_destroy(void* graph);

/* One horizontal band of the output (rows [row0,row0+rows) of the 2h output rows), computed from
 * the whole input frame resident on this device: the multi-GPU tiling of ONE large frame
 * (no counterpart in the reference; halo rows come from the source frame).
 * d_out_band receives rows*2w floats.  Bit-identical to the same rows of the whole-frame call. */
int srcnn_y_upscale2x_f32_band_dev(const float* d_in, unsigned w, unsigned h,
                                   unsigned row0, unsigned rows, float* d_out_band, void* stream);

/* ONE device-resident frame tiled over ALL contexts of this process (BASELINE config "single 8K frame tiled across 8
 * MI355X", from one process): d_in (w*h) and d_out (2w*2h) live on the calling thread's current context (the root).  Every
 * context pulls the source rows its output band needs from the root device, computes the band in `sub_bands` pieces
 * (<= 0: 4) and pushes each finished piece to d_out on the root with hipMemcpyPeerAsync while the next piece computes
 * (the push of sub-band k overlaps the kernels of k+1; one xGMI link per peer).  Synchronous: d_out is complete on return.
 * Bit-identical to srcnn_y_upscale2x_f32_dev.  The multi-PROCESS counterpart is srcnn_comm_tiled_y_upscale2x_f32_dev. */
int srcnn_y_upscale2x_f32_node_dev(const float* d_in, unsigned w, unsigned h, float* d_out, int sub_bands);

/* General Y path: resample to (dw,dh) with any SRCNNFilterType, then the three convolutions
 * (what doSRCNN does for non-2x factors / other filters; src/libsrcnn.cpp:662-723). */
int srcnn_y_path_f32_dev(const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh,
                         int filter, float* d_out, void* stream);

/* ---- per-kernel timing of the hot path (what bench.py's roofline object is computed from) ----
 * When enabled, every hot-path call brackets each of its kernels with HIP events recorded on the
 * SAME stream the kernel is launched on.  srcnn_profile_read synchronises on the recorded events and
 * returns, per stage, the accumulated device time and launch count since the last reset.
 * Stages: 0 = resampler (both passes), 1 = conv 9x9 + 1x1 (layers 1+2), 2 = conv 5x5 (layer 3). */
#define SRCNN_STAGE_RESAMPLE 0
#define SRCNN_STAGE_CONV12   1
#define SRCNN_STAGE_CONV3    2
#define SRCNN_STAGE_COUNT    3
int srcnn_profile_enable(int on);                       /* returns previous setting */
int srcnn_profile_reset(void);
int srcnn_profile_read(int stage, double* total_ms, unsigned long long* launches);   /* summed over the contexts */
int srcnn_profile_read_context(int context, int stage, double* total_ms, unsigned long long* launches);   /* one context */

/* ---- stage-level entry points (layer parity tests, debugging; each is one reference function) ---- */
/* FRAWResizeEngine::scale (src/frawscale.cpp:162-286) */
int srcnn_resample_f32_dev(const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh,
                           int filter, float* d_out, void* stream);
/* 64 x convolution99 (src/libsrcnn.cpp:350-422): d_out = 64 planes of w*h */
int srcnn_conv1_f32_dev(const float* d_y, unsigned w, unsigned h, float* d_c1, void* stream);
/* 32 x convolution11 (src/libsrcnn.cpp:424-447): d_c1 = 64 planes, d_c2 = 32 planes */
int srcnn_conv2_f32_dev(const float* d_c1, unsigned w, unsigned h, float* d_c2, void* stream);
/* convolution55 (src/libsrcnn.cpp:449-529): d_c2 = 32 planes, d_out = 1 plane */
int srcnn_conv3_f32_dev(const float* d_c2, unsigned w, unsigned h, float* d_out, void* stream);
/* fused 64xconvolution99 + 32xconvolution11 as used by the hot path: d_c2 = 32 planes */
int srcnn_conv12_f32_dev(const float* d_y, unsigned w, unsigned h, float* d_c2, void* stream);

/* ---- host-pointer conveniences (H2D, run, D2H, synchronous) ---- */
int srcnn_y_upscale2x_f32(const float* in, unsigned w, unsigned h, float* out);
int srcnn_y_upscale2x_f32_batch(const float* in, unsigned w, unsigned h, unsigned nframes, float* out);
int srcnn_y_path_f32(const float* in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, float* out);

/* Stream of frames in host memory (config "stream of 4K frames"): two slots of device frame buffers; all kernels on one
 * HIP stream (frames back to back), each slot's copies on its own copy stream, and -- when use_graph != 0 -- one hipGraph per
 * slot, captured once from the slot's kernel sequence and replayed per frame.  Copy/kernel dependencies are resolved on the
 * host (a helper thread queues a frame's D2H once its kernels are done): on this runtime a copy that waits device-side on
 * another queue does not overlap the kernels.  The caller's buffers are page-locked for the duration of the call unless they
 * already are (srcnn_host_alloc_pinned).  H2D of frame i+1 and D2H of frame i-1 overlap the kernels of frame i: 96 % of the
 * resident rate on 4K frames.  Identical results to nframes calls of srcnn_y_upscale2x_f32.
 * use_graph: 0 = plain launches (what a caller should ask for on ROCm 7.2: on this runtime a graph replay keeps a thread of
 * the RUNTIME spinning from launch to completion -- 9.9 ms of host CPU per 9.5 ms 4K frame against 0.6 ms, at the same
 * throughput); 1 = hipGraph replay KEPT ONLY WHILE IT IS CHEAP: the process's CPU time over the first four replayed frames is
 * held against their wall time, and above SRCNN_GRAPH_MAX_CPU_PCT (default 10 %: what plain launches cost) the graphs are retired and the stream
 * continues with plain launches (remembered per shape; srcnn_debug_stream_mode in srcnn_amd_debug.h says what ran);
 * 2 = hipGraph replay whatever it costs (BASELINE config "per-GPU hipGraph capture", measurements of the replay itself). */
int srcnn_y_upscale2x_f32_stream(const float* in, unsigned w, unsigned h, unsigned nframes, float* out, int use_graph);

/* One doSRCNN pass on an interleaved 8-bit RGB(A) image, fully on the device
 * (src/libsrcnn.cpp:628-923): colour split :233-272, per-plane resample :665-726, Y convolutions,
 * merge + clamp + truncate :274-308, optional truncated conv-Y :889-905.
 * out: (w*m)*(h*m)*d bytes, conv_opt: (w*m)*(h*m) bytes or NULL; both caller-allocated host memory.
 * Buffers that come from srcnn_host_alloc_pinned are copied from / into directly (no staging memcpy, no fan-out): what a
 * caller with a sequence of images should use (10.0 instead of 10.3-11.7 ms per 3840x2160 RGB image). */
int srcnn_process_u8(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply,
                     int filter, unsigned char* out, unsigned char* conv_opt);

/* The same call, asynchronous.  begin() validates nothing but `job`, starts the work and returns at once; wait() blocks until
 * the image is complete, returns what srcnn_process_u8 would have returned (its text in srcnn_last_error()) and frees the
 * job.  rgb / out / conv_opt must stay valid and untouched in between; every begin() needs exactly one wait().  A caller that
 * upscales a sequence keeps two jobs in flight, so that the device never idles between images (the harness of the reference
 * times one blocking call, src/test.cpp:653-672; this is for callers with more than one image). */
int srcnn_process_u8_begin(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply,
                           int filter, unsigned char* out, unsigned char* conv_opt, void** job);
int srcnn_process_u8_wait(void* job);

/* Output geometry of ProcessSRCNN for (w,h,multiply) with or without step scaling: the reference only
 * returns a byte count (outbuffsz) and truncates w*m pass by pass (src/libsrcnn.cpp:662-663, 980-1061). */
int srcnn_output_size(unsigned w, unsigned h, float multiply, int stepscale, unsigned* out_w, unsigned* out_h);

/* delete[] for buffers handed out by ProcessSRCNN (outbuff / *convbuff), for callers that cannot
 * run C++ delete[] themselves (ctypes, cgo, JNI ...).  The reference leaves this to the caller. */
void srcnn_delete_array(unsigned char* p);

/* ---- multi-GPU, one process per GPU: RCCL over xGMI only for the band gather.  The communicator binds to the calling
 * thread's current context (its device).
 * The unique id is produced on rank 0 and handed to the other ranks by the caller's own
 * bootstrap (torch.distributed/gloo store, MPI, a file ...). */
#define SRCNN_COMM_ID_BYTES 128
int srcnn_comm_unique_id(unsigned char id[SRCNN_COMM_ID_BYTES]);
int srcnn_comm_init(const unsigned char id[SRCNN_COMM_ID_BYTES], int rank, int nranks);
int srcnn_comm_destroy(void);
int srcnn_comm_rank(int* rank, int* nranks);
/* every rank contributes `count` floats at d_send; rank `root` receives nranks*count floats in
 * rank order at d_recv (ignored elsewhere).  Direct peer->root sends, one xGMI link each. */
int srcnn_comm_gather_f32(const float* d_send, size_t count, float* d_recv, int root, void* stream);
/* The same with one count per rank (counts[nranks], identical on every rank): rank r's counts[r] floats land at
 * d_recv + counts[0] + ... + counts[r-1] on the root, so bands of unequal height -- an output height the rank
 * count does not divide -- assemble into one contiguous frame.  A count may be 0. */
int srcnn_comm_gatherv_f32(const float* d_send, const size_t* counts, float* d_recv, int root, void* stream);
/* The same with explicit destinations: rank r's counts[r] floats land at d_recv + offsets[r] on the root (counts and
 * offsets identical on every rank).  This is what lets a band be gathered piece by piece while it is being computed. */
int srcnn_comm_gatherv_at_f32(const float* d_send, const size_t* counts, const size_t* offsets, float* d_recv, int root,
                              void* stream);
/* ONE frame tiled over the ranks of the communicator, compute and gather overlapped (BASELINE config "single 8K frame tiled
 * across 8 MI355X ... RCCL gather over xGMI"): this rank's band of the 2h output rows (rows split as evenly as possible,
 * earlier ranks take the remainder) is computed in `sub_bands` pieces (<= 0: 4) on `stream`; as soon as piece k's kernels
 * are queued, its gather (peer->root ncclSend/ncclRecv, one group) is queued on the library's comm stream behind an event,
 * so it runs while piece k+1 computes.  d_in: the whole w*h source frame on every rank; d_band: scratch for this rank's
 * band (rows*2w floats); d_full: the 2w*2h result on the root (ignored elsewhere).  On return everything is queued and
 * `stream` has been made to wait for the last gather: synchronise `stream` to use d_full.  Every rank must call it with the
 * same w, h, root and sub_bands.  Bit-identical to srcnn_y_upscale2x_f32_dev on the root. */
int srcnn_comm_tiled_y_upscale2x_f32_dev(const float* d_in, unsigned w, unsigned h, float* d_band, float* d_full, int root,
                                         int sub_bands, void* stream);
/* rows [*row0, *row0 + *rows) of the out_h output rows that `rank` of `nranks` owns in the tiled path above */
int srcnn_band_rows(unsigned out_h, int rank, int nranks, unsigned* row0, unsigned* rows);
/* piece `piece` of `npieces` of that band of an (out_w x out_h) frame (what one sub-band gather of the tiled path moves):
 * large pieces first, a short one last, each cut where it fills whole rounds of the persistent layer-1+2 grid; a pure function
 * of its arguments (no device needed), identical on every rank.  A short band yields fewer pieces: the rest have rows == 0. */
int srcnn_tiled_piece(unsigned out_w, unsigned out_h, int rank, int nranks, int piece, int npieces, unsigned* row0, unsigned* rows);
int srcnn_comm_allgather_f32(const float* d_send, size_t count, float* d_recv, void* stream);
int srcnn_comm_barrier(void* stream);
/* No wait on a peer is unbounded.  Every RCCL call that can block on the host runs under a deadline, and the host waits for
 * queued communication with srcnn_comm_wait (a bounded srcnn_stream_sync; srcnn_comm_barrier uses it).  When the deadline
 * passes -- a rank died, or the ranks derived different gather tables -- the communicator is aborted (ncclCommAbort), the
 * call returns SRCNN_E_COMM, and every later srcnn_comm_* call fails at once until srcnn_comm_destroy + srcnn_comm_init.
 * Default 60000 ms, env SRCNN_COMM_TIMEOUT_MS; 0 = no deadline.  srcnn_comm_set_timeout_ms returns the previous value.
 * srcnn_comm_destroy drains what is still queued under the same deadline and aborts instead of destroying on a miss.
 * The first time a rank uses a counts / offsets table, a checksum of the table is all-reduced (16 bytes, the host waits for
 * it: do not make that first call inside a stream capture), so that ranks which disagree FROM THE START return SRCNN_E_COMM
 * together, in milliseconds, instead of pairing a send with the wrong receive (SRCNN_COMM_CHECK=0 skips it).  A rank verifies
 * a table once; ranks that agreed on one table and disagree later (one still has its table verified, the other arrives with
 * a new one) cannot be paired by the check and are caught by the deadline instead: every rank returns SRCNN_E_COMM after
 * SRCNN_COMM_TIMEOUT_MS, the communicator is aborted.  The tables the library derives itself depend on (width, height, ranks,
 * pieces) only -- never on a switch.
 * srcnn_comm_destroy waits, under the same deadline, for everything that was queued through srcnn_comm_*: it records an event
 * behind the last communication on every stream it was given -- a raw HIP stream of the caller's included, and an event stays
 * valid after its stream is destroyed -- and aborts instead of destroying the communicator when one does not complete. */
int srcnn_comm_wait(void* stream);
int srcnn_comm_set_timeout_ms(int ms);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SRCNN_AMD_H */
