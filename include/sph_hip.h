/*
 * sph_hip.h -- C ABI of libsph_hip, the MI355X (gfx950) engine for the SPH frame step.
 *
 * Every entry point replaces one use of UnityEngine.ComputeShader / ComputeBuffer in the
 * reference host driver Assets/SPH.cs (citations are "SPH.cs:line") or one kernel of
 * Assets/SPHCompute.compute ("compute:line").  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.
 *
 * Conventions
 *   - every function returns int: 0 = SPH_OK, <0 = SPH_E_*; nothing throws or aborts.
 *   - one ctx = one GPU = one owner thread at a time.  Calls enqueue on the ctx's HIP
 *     stream and return; only download / read_counter / get_timings / synchronize block.
 *   - the library owns all device memory; the caller owns every host pointer it passes.
 *   - particle records at the boundary are the reference's 72-byte AoS struct
 *     (compute:26-37, SPH.cs:54-65): 18 x fp32
 *       pos[0..2] vel[3..5] acc[6..8] density[9] pressure[10] restDensity[11]
 *       viscosity[12] mass[13] color[14..17]
 *     always in ORIGINAL particle-id order, whatever order the engine keeps internally.
 */
#ifndef SPH_HIP_H
#define SPH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_OK              0
#define SPH_E_INVALID      -1   /* bad argument / bad state (message in sph_last_error) */
#define SPH_E_NOMEM        -2   /* device or host allocation failed                     */
#define SPH_E_HIP          -3   /* a HIP runtime call failed                            */
#define SPH_E_NO_DEVICE    -4   /* no usable gfx950 device                              */
#define SPH_E_STATE        -5   /* call order violated (e.g. density before grid build) */

#define SPH_PARTICLE_FLOATS 18
#define SPH_PARTICLE_BYTES  72

/* Kernel ids = "#pragma kernel" order of the reference (compute:9-17); these are the
 * values ComputeShader.FindKernel returns in SPH.cs:366-377. */
enum sph_kernel {
    SPH_K_CLEAR            = 0,  /* CS_Clear            compute:87-98   */
    SPH_K_CLEAR_GRID       = 1,  /* CS_ClearGrid        compute:104-111 */
    SPH_K_BUILD_GRID       = 2,  /* CS_BuildGrid        compute:117-136 */
    SPH_K_DENSITY_PRESSURE = 3,  /* CS_DensityPressure  compute:142-193 */
    SPH_K_FORCE_XSPH       = 4,  /* CS_ForceXSPH        compute:199-303 */
    SPH_K_VV_HALF_STEP     = 5,  /* CS_VvHalfStep       compute:309-319 */
    SPH_K_VV_FULL_STEP     = 6,  /* CS_VvFullStep       compute:325-334 */
    SPH_K_BOUND_OBS        = 7,  /* CS_BoundObs         compute:340-486 */
    SPH_K_VELOCITY_DAMPING = 8,  /* CS_VelocityDamping  compute:492-509 */
    SPH_K_COUNT            = 9
};

/* Field-for-field mirror of "cbuffer SimulationParams" (compute:47-81), in declaration
 * order, as filled by SPH.SetComputeParams (SPH.cs:379-412), UpdateObstacle (348-361) and
 * the _DeltaTime upload (208).  Matrices are row-major M[r*4+c] and act on column vectors,
 * i.e. mul(M, v)[r] = sum_c M[r*4+c] * v[c]  (compute:351, 396, 408). */
typedef struct sph_params {
    uint32_t struct_size;               /* = sizeof(sph_params); versioning            */
    int32_t  particleCount;             /* _ParticleCount                              */
    float    deltaTime;                 /* _DeltaTime                                  */
    float    particleMass;              /* _ParticleMass   (declared, read by no kernel) */
    float    restDensity;               /* _RestDensity    (declared, read by no kernel) */
    float    soundSpeed;                /* _SoundSpeed                                 */
    float    gamma;                     /* _Gamma                                      */
    float    smoothingRadius;           /* _SmoothingRadius                            */
    float    gravity;                   /* _Gravity                                    */
    float    surfaceTensionCoefficient; /* _SurfaceTensionCoefficient                  */
    float    surfaceTensionThreshold;   /* _SurfaceTensionThreshold                    */
    float    xsphEpsilon;               /* _XSPHEpsilon                                */
    float    particleCollisionDamping;  /* _ParticleCollisionDamping                   */
    float    obstacleRepulsionStiffness;/* _ObstacleRepulsionStiffness (read by no kernel) */
    float    obstacleRadius;            /* _ObstacleRadius                             */
    float    boundsCenter[3];           /* _BoundsCenter   (AABB branch only)          */
    float    boundsSize[3];             /* _BoundsSize     (AABB branch only)          */
    float    spawnCenter[3];            /* _SpawnCenter    (read by no kernel)         */
    float    spawnSize[3];              /* _SpawnSize      (read by no kernel)         */
    float    obstaclePos[3];            /* _ObstaclePos                                */
    int32_t  gridResolutionX;           /* _GridResolutionX                            */
    int32_t  gridResolutionY;           /* _GridResolutionY                            */
    int32_t  gridResolutionZ;           /* _GridResolutionZ                            */
    int32_t  maxParticlesPerCell;       /* _MaxParticlesPerCell                        */
    float    cellSize;                  /* _CellSize                                   */
    float    minBound[3];               /* _MinBound                                   */
    float    boundaryMatrix[16];        /* _BoundaryMatrix    (cube localToWorld)      */
    float    boundaryInvMatrix[16];     /* _BoundaryInvMatrix (cube worldToLocal)      */
    float    boundaryHalfExtents[3];    /* _BoundaryHalfExtents: only |.|>0.001 is read,
                                           as the OBB/AABB mode flag (compute:349)     */
} sph_params;

/* One spawn box's static per-particle properties (SpawnBox.cs:10-23); the device-resident
 * stand-in for the arrays SPH.UpdateParticleStaticProperties rewrites (SPH.cs:284-298). */
typedef struct sph_phase {
    float restDensity;
    float viscosity;
    float mass;
    float color[4];
} sph_phase;

/* Soft-fault / statistics words (never errors). */
enum sph_counter {
    SPH_CTR_COLLISIONS   = 0, /* CollisionCounter[0] (compute:418); zeroed by sph_reset_counter */
    SPH_CTR_BIN_OVERFLOW = 1, /* #cells whose population exceeded maxParticlesPerCell at the
                                 last grid build (compute:131 drops those ids)            */
    SPH_CTR_OUT_OF_GRID  = 2, /* #particles whose unclamped cell was outside the grid at
                                 the last grid build (compute:126 clamps them)            */
    SPH_CTR_MAX_PLANE_SHIFT = 3, /* slab mode: max |search plane - binned plane| over the local
                                 particles in the last CS_DensityPressure (halo-width check)     */
    SPH_CTR_BAND_MISSES  = 4, /* slab mode with sph_slab_arm_pack: owned particles that ended the last frame in a
                                 plane of the send range without being in a band workgroup, i.e. that were NOT
                                 sent (can only happen when the halo condition is violated too); counted by
                                 the grid build that follows                                                  */
    SPH_CTR_NAN          = 5, /* failure detection: particles whose position or velocity was not finite when a
                                 boundary pass (CS_BoundObs, or the fused tail of sph_step_frame) wrote them, since
                                 the last grid build; a particle that stays non-finite counts once per sub-step.
                                 Slab mode: owned particles only.  Never an error: the frame goes on, like the
                                 reference's would                                                              */
    SPH_CTR_COUNT        = 6
};

typedef struct sph_ctx sph_ctx;

/* Per-stage device time in milliseconds, AVERAGED over the timed sph_step_frame calls since the previous
 * sph_get_timings call (which resets the average).  The times are those of the dispatch packets themselves: start / stop
 * events attached to the frame's own launches (hipExtLaunchKernelGGL), so a timed frame enqueues exactly what an untimed
 * one does (a hipEventRecord between kernels is a barrier packet of its own, ~5 us of stream time each).
 * build_grid_ms runs from the first launch of the grid build to the end of the reorder (which also does the first half
 * step); full_step_ms is the fused tail (full step + boundary + damping); frame_ms from the first launch to the last;
 * half_step_ms / bound_obs_ms / damping_ms are 0 (those stages are fused).  With subSteps > 1 the sub-step stages are
 * those of the last sub-step.  SPH_HIP_TIMING=0 records nothing, 2 times every frame; SPH_HIP_TIMING_STRIDE=k times
 * every k-th frame (default 2: a launch that carries events costs ~70 us of host time, and with five of them in EVERY
 * frame a 0.4 ms frame loop becomes host-bound).  `frames` is the number of frames that were timed. */
typedef struct sph_timings {
    float build_grid_ms;   /* hash + counting sort by cell + reorder             */
    float half_step_ms;
    float density_ms;
    float force_ms;
    float full_step_ms;    /* full step (+ bound/obstacle when fused)            */
    float bound_obs_ms;
    float damping_ms;
    float frame_ms;        /* whole frame, first launch to last                   */
    int32_t frames;        /* number of frames averaged                           */
} sph_timings;

/* ---- lifetime ------------------------------------------------------------------------ */
/* Replaces the implicit device of the Unity player.  device_id = HIP ordinal.
 * One context drives ONE device: more GPUs are more contexts, one per process and GPU (the sph_slab_* entry points
 * below make a context one slab of a decomposed run, the host exchanges the face messages -- sph_amd/slab.py over
 * RCCL).  A context spanning several devices in one process, as a (ctx**, const int* device_ids, int n) signature
 * would suggest, does not exist on purpose: the messages between slabs belong to the host's communicator. */
int  sph_create(sph_ctx** out, int device_id);
/* SPH.OnDestroy (SPH.cs:235-241): ComputeBuffer.Release x4. */
int  sph_destroy(sph_ctx* ctx);
const char* sph_last_error(const sph_ctx* ctx);
/* Human-readable build info ("gfx950 ..."); never NULL. */
const char* sph_version(void);

/* ---- particle buffer ("new ComputeBuffer(N, 72)", SetData, GetData) ----------------------- */
/* SPH.cs:248-249  new ComputeBuffer(particleCount, 18*4). */
int  sph_alloc_particles(sph_ctx* ctx, int n);
/* SPH.cs:280 / 297  particleBuffer.SetData(array): n records of 72 B in id order. */
int  sph_upload_particles(sph_ctx* ctx, const void* aos72, int n);
/* particleBuffer.SetData(array) in the MIDDLE of a frame: the reference's SetData never touches GridCounts /
 * GridIndices (SPH.cs:280, 297 vs 156-157), so the records are written over the particles where they sit and the grid
 * of the last CS_BuildGrid stays valid.  What lock-step tests use to re-synchronise between sub-steps. */
int  sph_upload_particles_in_place(sph_ctx* ctx, const void* aos72, int n);
/* SPH.cs:287  particleBuffer.GetData(array): n records of 72 B, id order.  Synchronises. */
int  sph_download_particles(sph_ctx* ctx, void* aos72, int n);
/* Render feed.  In the reference the renderer binds the simulation's ComputeBuffer itself
 * (SPH.cs:166-170 material.SetBuffer("_ParticleBuffer", particleBuffer); ParticleRender.shader:54-67 reads the
 * 72-byte records, id = SV_VertexID).  Here: the same n records, id order, written into a DEVICE buffer of the
 * caller on the ctx's GPU (n * 72 bytes; checked against the extent of its allocation), enqueued on the ctx's stream -- no host copy, does not
 * synchronise.  Not available in slab mode. */
int  sph_export_particles_device(sph_ctx* ctx, void* dev_aos72, int n);

/* ---- grid buffers ---------------------------------------------------------------------- */
/* SPH.cs:156-157 / 339-345: (re)allocation of GridCounts[cells] and GridIndices[cells*cap].
 * The engine keeps one cumulative cell table (a counting sort by cell) instead (no cells*cap array). */
int  sph_resize_grid(sph_ctx* ctx, int rx, int ry, int rz, int max_per_cell);

/* ---- parameters ------------------------------------------------------------------------ */
/* SPH.cs:379-412 SetComputeParams + 348-361 UpdateObstacle + 208 SetFloat("_DeltaTime"). */
int  sph_set_params(sph_ctx* ctx, const sph_params* p);
/* SPH.cs:208  sphCompute.SetFloat("_DeltaTime", dtSub) on its own. */
int  sph_set_delta_time(sph_ctx* ctx, float dt);

/* ---- static per-particle properties (SPH.cs:284-298 without the PCIe round trip) -------- */
int  sph_set_phase_table(sph_ctx* ctx, const sph_phase* table, int n_phases);
/* particleSpawnBoxIndices (SPH.cs:75, 276): one phase index per particle id. */
int  sph_set_phase_ids(sph_ctx* ctx, const uint8_t* ids, int n);
/* Applies table[phase[id]] to restDensity/viscosity/mass/color of every particle now
 * (what UpdateParticleStaticProperties does each frame).  Slab mode: the phase is kept per slot (sph_slab_record.phase)
 * and a table in force is authoritative, as the spawn boxes are in the reference -- the local slots (owned and ghost)
 * take their rows here, records stored later (sph_slab_upload, sph_slab_rebuild) take theirs when they are stored, so a
 * live edit made on every rank reaches every particle; a phase beyond the table keeps the record's own values. */
int  sph_apply_phase_table(sph_ctx* ctx);

/* ---- dispatch -------------------------------------------------------------------------- */
/* SPH.DispatchCompute(kernel, count) (SPH.cs:439-443).  count must be the particle count
 * (or the cell count for SPH_K_CLEAR_GRID), as SPH.cs always passes. */
int  sph_dispatch(sph_ctx* ctx, int kernel_id, int count);
/* The body of SPH.Update from ClearGrid to VelocityDamping (SPH.cs:201-222) in one call,
 * with the same stage order; sets _DeltaTime = timeStep / subSteps itself (SPH.cs:207-208). */
int  sph_step_frame(sph_ctx* ctx, int sub_steps, float time_step);
/* Block until everything enqueued on the ctx's stream has finished. */
int  sph_synchronize(sph_ctx* ctx);

/* ---- counters (CollisionCounter, SPH.cs:158, 190) ---------------------------------------- */
int  sph_reset_counter(sph_ctx* ctx, int which);            /* SetData({0}) SPH.cs:190 */
int  sph_read_counter(sph_ctx* ctx, int which, int* value); /* synchronises            */

/* ---- parity / debug exports (no reference counterpart; used by tests and the harness) ---- */
/* State of the last grid build.  Any pointer may be NULL.
 *   cell_start/cell_end[cells] : range of the cell in sorted order (equal when empty);
 *                                 cell_end-cell_start = GridCounts[cell] (compute:130)
 *   sorted_ids[n]               : particle ids by (cell, id): the concatenation of the
 *                                 reference's bins when each is filled in ascending id
 *   cell_of_particle[n]         : clamped linear cell index per particle id (compute:125-127) */
int  sph_download_grid(sph_ctx* ctx, int32_t* cell_start, int32_t* cell_end,
                       int32_t* sorted_ids, uint32_t* cell_of_particle);
/* Neighbour lists as CS_DensityPressure would visit them NOW (current positions, current
 * grid): for particle id i, the ids j != i with rSq < h2 in visiting order (z,y,x cells,
 * then bin order; compute:158-181).  counts[n]; lists[n*max_neighbors], entries beyond
 * counts[i] are set to -1; counts[i] may exceed max_neighbors (then the list is truncated). */
int  sph_download_neighbors(sph_ctx* ctx, int32_t* counts, int32_t* lists, int max_neighbors);
/* The hit lists CS_ForceXSPH is about to walk: the bit mask CS_DensityPressure's engine kernel left for the force kernel
 * (one bit per candidate slot, the reference's loops compute:170-181 / 241-246 evaluated once instead of twice), decoded
 * into ids in visiting order -- the artefact the force kernel actually consumes, where sph_download_neighbors re-walks
 * the candidates.  Same layout as sph_download_neighbors; counts[i] = -1 for a particle without a usable mask (its
 * rows need more words than a mask row holds; its wave scans instead).  *mask_state: 1 = lists written, 0 = no fresh
 * mask exists (positions or grid changed since the last CS_DensityPressure: CS_ForceXSPH would scan), 2 = a bin
 * overflowed its capacity at the last grid build (rule O2: nothing is recorded, the capped 27-cell scan runs);
 * for 0 and 2 counts / lists are not touched.  Not available in slab mode.  Synchronises. */
int  sph_download_hit_lists(sph_ctx* ctx, int32_t* counts, int32_t* lists, int max_neighbors, int* mask_state);
int  sph_get_timings(sph_ctx* ctx, sph_timings* out);
/* Raw device pointers of the internal SoA buffers, for zero-copy views (torch, RCCL).
 * name: "posM" "velV" "acc" "rhoP" "rho0" "origId".  Valid until the next alloc/destroy. */
int  sph_device_pointer(sph_ctx* ctx, const char* name, void** ptr, uint64_t* bytes);
/* The HIP stream the ctx enqueues on (hipStream_t as void*). */
int  sph_get_stream(sph_ctx* ctx, void** stream);

/* ---- multi-GPU slab decomposition (no reference counterpart: the reference is single-GPU) ----------
 * One ctx per GPU holds the particles of an x-slab of whole cell planes of the GLOBAL grid plus ghost
 * copies of the neighbouring planes.  sph_params.gridResolutionX stays the GLOBAL x resolution;
 * sph_resize_grid gets the LOCAL one (owned + ghost planes).  Particle ids are global.  See
 * sph_amd/slab.py for the per-frame protocol (one message per slab face per frame). */
typedef struct sph_slab_record {       /* 64 bytes: what travels between slabs */
    float    pos[3], mass;
    float    vel[3], viscosity;
    float    acc[3], restDensity;
    uint32_t id;                        /* global particle id */
    float    density, pressure;
    uint32_t phase;
} sph_slab_record;

/* Local grid = global planes [x_offset, x_offset + rx_local) (rx_local from sph_resize_grid); planes
 * [own_lo, own_hi) (GLOBAL coordinates) are owned.  rx_global <= 0 switches slab mode off.  May be called again
 * between frames to move the cut (see sph_slab_plane_counts). */
int  sph_slab_configure(sph_ctx* ctx, int rx_global, int x_offset, int own_lo, int own_hi);
/* Capacity-based allocation for a slab whose population changes every frame. */
int  sph_slab_reserve(sph_ctx* ctx, int capacity);
/* Replace the local set by n records from host memory (initial distribution). */
int  sph_slab_upload(sph_ctx* ctx, const sph_slab_record* records, int n);
/* Local particles (owned only, or owned + ghosts) as records, to host memory.  Synchronises. */
int  sph_slab_download(sph_ctx* ctx, sph_slab_record* records, int max_records, int owned_only, int* n);
/* Pack the face messages into DEVICE buffers: owned particles whose current global plane is
 * < left_below go to dev_left, >= right_from to dev_right.  Counts may exceed the capacities
 * (then the message is truncated and the call fails).  Synchronises. */
int  sph_slab_pack(sph_ctx* ctx, int left_below, int right_from, void* dev_left, int cap_left,
                   void* dev_right, int cap_right, int* n_left, int* n_right);
/* New local set = owned particles whose current global plane is in [keep_lo, keep_hi) + the received
 * records (DEVICE buffers) in that plane range; last frame's ghosts are dropped.  The call only appends
 * the records behind the current particles and arms a filter: the NEXT grid build (CS_BuildGrid
 * dispatch or sph_step_frame), which must follow before anything else reads the particles, drops
 * what does not belong while it sorts -- no separate compaction pass.  *n_local is therefore an upper
 * bound (what CS_ClearGrid / CS_BuildGrid dispatches must pass as count); sph_particle_count gives
 * the exact number after the build.  Does not synchronise. */
int  sph_slab_rebuild(sph_ctx* ctx, int keep_lo, int keep_hi, const void* dev_from_left, int n_from_left,
                      const void* dev_from_right, int n_from_right, int* n_local);
/* Exchange overlapped with compute.  sph_slab_arm_pack makes the NEXT sph_step_frame pack that frame's face
 * messages itself (same selection and record contents as sph_slab_pack called after the frame), as early as
 * the frame allows: the last sub-step's CS_ForceXSPH runs first on the workgroups (256 consecutive slots)
 * that hold an owned particle binned within band_margin planes of the send range, the records are written
 * with the rest of the frame applied on the fly, and only then the remaining workgroups run.  sph_step_frame
 * returns as usual; sph_slab_pack_wait blocks until the records are complete -- while the GPU continues with the
 * rest of the frame -- and returns the counts.  The caller then moves the buffers and calls sph_slab_rebuild,
 * which is enqueued behind the frame.  band_margin must cover the planes a particle can move in one frame
 * (ghost/2 does whenever the halo condition holds); particles that were missed are counted by the following
 * grid build (SPH_CTR_BAND_MISSES).  left_below = 0 / right_from = global resolution: no neighbour on that side. */
int  sph_slab_arm_pack(sph_ctx* ctx, int left_below, int right_from, int band_margin, void* dev_left, int cap_left,
                       void* dev_right, int cap_right);
/* n_band_workgroups (may be NULL): how many 256-slot workgroups ran before the pack (of ceil(n / 256)). */
int  sph_slab_pack_wait(sph_ctx* ctx, int* n_left, int* n_right, int* n_band_workgroups);
/* Load balance: particles per LOCAL cell plane (owned and ghost planes alike) as binned by the last grid build,
 * counts[n_planes] with n_planes = the local x resolution.  Read from the cell table (no particle pass); needs
 * the grid of the last frame, i.e. call it before sph_slab_rebuild.  Synchronises.
 * Moving a slab's cut: sph_slab_configure + sph_resize_grid with the new ranges BEFORE that frame's
 * sph_slab_pack; ownership (who sends what) stays that of the last grid build until the next one, so a frame's
 * messages carry whatever changes hands.  The driver must keep every new face, plus the ghost width, inside the
 * two slabs that were adjacent to the old face (sph_amd/slab.py: SlabEngine.rebalance). */
int  sph_slab_plane_counts(sph_ctx* ctx, int32_t* counts, int n_planes);
/* sph_slab_plane_counts for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same result as sph_slab_plane_counts (also available
 * outside slab mode: n_planes = the grid's x resolution), but the per-plane counters the kernel accumulates into live
 * in a static device array of the library (SPH_PLANE_COUNTS_MAX ints = 32 KiB per GPU, held until the process ends)
 * instead of a buffer allocated on demand: the call makes no hipMalloc / hipFree.  n_planes > SPH_PLANE_COUNTS_MAX is
 * rejected (SPH_E_INVALID).  `counts` may be device memory of the ctx's GPU or host memory (pinned host memory if the
 * caller also forbids the runtime's own staging of pageable copies).  Every call on a GPU shares that GPU's array; the
 * library serialises them (a per-device lock held for the duration of the call, which synchronises anyway), so two
 * ctxs on one device may call it from two threads.  Synchronises. */
#define SPH_PLANE_COUNTS_MAX 8192
int  sph_slab_plane_counts_noalloc(sph_ctx* ctx, int32_t* counts, int n_planes);
/* Current local particle count (owned + ghosts). */
int  sph_particle_count(sph_ctx* ctx, int* n);

#ifdef __cplusplus
}
#endif
#endif /* SPH_HIP_H */
