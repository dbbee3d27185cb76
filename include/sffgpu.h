/* sffgpu.h — C ABI of the MI355X-native SFF / SFF* hot path (libsffgpu.so).
 *
 * Drop-in boundary for the tree-expansion loop of ctu-mrs/space_filling_forest_star:
 * each entry point replaces one call the reference makes into FLANN / RAPID / its own
 * Solver base (file:line of the reference interface given per function).  Plain pointers
 * and sizes only; all buffers are caller-owned HOST memory unless a name ends in `_dev`.
 * Every call returns 0 on success and a negative code on failure, with a message
 * available from sffgpu_last_error().  A context is bound to one GPU and is not
 * thread-safe (the reference is single-threaded and keeps state in statics:
 * src/primitives.h:443-445,493, src/environment.h:76).  There is NO CPU fallback: without a
 * usable gfx950 device sffgpu_create() fails.
 */
#ifndef SFFGPU_H
#define SFFGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sffgpu_ctx sffgpu_ctx;
typedef struct sffgpu_forest sffgpu_forest;

enum { SFFGPU_OK = 0, SFFGPU_ERR_ARG = -1, SFFGPU_ERR_HIP = -2, SFFGPU_ERR_STATE = -3, SFFGPU_ERR_CAPACITY = -4 };
enum { SFFGPU_MESH_ENV = 0, SFFGPU_MESH_ROBOT = 1 };

/* library / device */
const char* sffgpu_version(void);
int sffgpu_device_count(void);
int sffgpu_create(int device, sffgpu_ctx** out);
void sffgpu_destroy(sffgpu_ctx* ctx);
const char* sffgpu_last_error(sffgpu_ctx* ctx); /* ctx may be NULL: last create() error */

/* Collision models.  Replaces RAPID_model::BeginModel/AddTri/EndModel as driven by
 * Obstacle<T>::Obstacle / addFacet (src/environment.h:101-115, :212-223).  tri9 = n x 9 doubles
 * (three xyz vertices), already offset and scaled like the reference parser does.  All obstacles
 * of an Environment are merged into the one ENV model (Environment::Collide ORs over them,
 * src/environment.h:306-316).  The ENV model may be empty (HasMap == false). */
int sffgpu_mesh_upload(sffgpu_ctx* ctx, int role, const double* tri9, int n_tri);

/* Environment::Collide(position) (src/environment.h:306-316 -> RAPID_Collide :268-276):
 * hit[i] = 1 if the robot posed at pos6[i] = (x y z yaw pitch roll) touches any ENV triangle. */
int sffgpu_collide_poses(sffgpu_ctx* ctx, const double* pos6, int n, uint8_t* hit);

/* The same with the robot placed by an explicit rigid transform, x_world = R x_model + T: rt12[i] = the 3x3
 * rotation (row-major) followed by the translation.  This is RAPID_Collide's own calling convention
 * (src/environment.h:246,274 pass rotation matrices and translation vectors) and what include/sff/RAPID.H forwards to. */
int sffgpu_collide_transforms(sffgpu_ctx* ctx, const double* rt12, int n, uint8_t* hit);

/* Solver::isPathFree(start, finish) (src/problemStruct.h:154-168), batched over n edges.
 * is_free[i] = 1 if no interpolated sample collides; first_hit[i] = index of the first colliding
 * sample (the reference stops there) or -1; n_samples[i] = samples the edge has.  The last two
 * may be NULL. */
int sffgpu_collide_segments(sffgpu_ctx* ctx, const double* a6, const double* b6, int n, uint8_t* is_free,
                            int32_t* first_hit, int32_t* n_samples);

/* RandGen::randomPointInDistance (src/randGen.h:70-109) for n centres in one launch.  `words`
 * are raw std::mt19937_64 outputs in the reference's draw order: 6 per sample when dim == 6,
 * 1 when dim == 2 (always strided by 6).  limits = minX maxX minY maxY minZ maxZ. */
int sffgpu_sample_steer(sffgpu_ctx* ctx, const uint64_t* words, const double* center6, int n, double dist, int dim,
                        const double limits[6], double* out6, uint8_t* in_limits);

/* Node store = the FLANN index of every tree (Tree::flannIndex, src/primitives.h:506).
 * append replaces Index::addPoints (src/forest.h:367, src/rrt.h:215); tree_id tags each node. */
int sffgpu_nodes_reset(sffgpu_ctx* ctx, int capacity);
int sffgpu_nodes_append(sffgpu_ctx* ctx, const double* pos6, const int32_t* tree_id, int n);
int sffgpu_nodes_count(sffgpu_ctx* ctx);
/* Device time of the context's kernels since it was created, measured with HIP events on the library's launch
 * stream: [0] neighbour query (sweep / grid), [1] collision, [2] sampling; launches = kernel groups timed. */
int sffgpu_kernel_times(sffgpu_ctx* ctx, double ms[3], uint64_t launches[3]);

/* Exact radius query, replaces Index::radiusSearch (src/forest.h:266-267).  For each of nq
 * queries returns every stored node with 6-D distance < r[q] (true metric of
 * src/primitives.h:224-235, double), restricted to tree[q] (or all trees when tree[q] < 0) and
 * to node ids < max_id[q] (all when max_id is NULL).  Results per query are sorted by
 * (distance, node id); idx/dist are nq x cap, cnt[q] is the TOTAL number found (entries beyond
 * cap are dropped, check cnt[q] <= cap). */
int sffgpu_radius(sffgpu_ctx* ctx, const double* q6, int nq, const double* r, const int32_t* tree,
                  const int32_t* max_id, int32_t* idx, double* dist, int32_t* cnt, int cap);

/* Exact k nearest, replaces Index::knnSearch (src/forest.h:317, src/rrt.h:143,166,228); same
 * filters and ordering as sffgpu_radius; cnt[q] = min(k, eligible nodes). idx/dist are nq x k. */
int sffgpu_knn(sffgpu_ctx* ctx, const double* q6, int nq, int k, const int32_t* tree, const int32_t* max_id,
               int32_t* idx, double* dist, int32_t* cnt);

/* Spatial index over the node store, the counterpart of Index::buildIndex (src/forest.h:66-72, src/rrt.h:56-62): a
 * uniform grid over the xyz limits with cells of edge `cell` (>= the typical query radius; it re-cells itself as nodes
 * get denser).  Nodes appended later enter it in the same launch.  With the index, sffgpu_knn calls that are not
 * restricted to one tree answer every query from the cells around it (k_knn_grid: time independent of the store's size)
 * instead of sweeping the whole store per query (k_knn_linear); the results are the same exact lists.
 * sffgpu_nodes_reset drops it. */
int sffgpu_nodes_index(sffgpu_ctx* ctx, const double limits[6], double cell);

/* ---------------------------------------------------------------- solver session
 * SpaceForest<T,R> (src/forest.h:31-54): constructor :57-110, Solve() main loop :113-202,
 * expandNode :240-376, maxConnected :379-418 — run as waves of `wave` frontier slots evaluated
 * together on the GPU; wave == 1 is the reference's sequential loop. */
typedef struct {
  int32_t dim;              /* Dimensions: 2 (D2) or 6 (D3), src/primitives.h:75-78 */
  int32_t optimize;         /* Problem::optimal -> SFF* choose-parent + rewire */
  int32_t has_goal;         /* Problem::hasGoal */
  double goal[6];           /* Problem::goal (already scaled) */
  double limits[6];         /* Environment::limits */
  double dist_tree;         /* Problem::distTree (scaled) */
  double sampling_dist;     /* Node::SamplingDistance (scaled) */
  int32_t threshold_misses; /* Node::ThresholdMisses */
  int32_t max_iterations;   /* Problem::maxIterations */
  int32_t node_budget;      /* extra stop: total nodes >= budget (0 = off; not in the reference) */
  int32_t wave;             /* frontier slots per wave */
  uint64_t seed;            /* mt19937_64 seed (reference: clock, src/randGen.h:52-55) */
  int32_t rank;             /* multi-GPU: this process's shard of every wave ... */
  int32_t world;            /* ... out of `world` shards (1 = single GPU) */
  double priority_bias;     /* Problem::priorityBias: != 0 selects frontier nodes through the priority heaps of
                               src/heap.h (best node with this probability, a random one otherwise; src/forest.h:126-147) */
  int32_t libm_sampling;    /* 1 = parity mode: RandGen::randomPointInDistance (src/randGen.h:70-109) is evaluated on the
                               host with the C library's cos / sin / acos - the reference's own arithmetic - and the
                               samples are uploaded; 0 = sampled on the GPU with the portable trig of csrc/sff_pmath.h
                               (<= 1 ulp from glibc).  Slower (host engine, one extra upload per round); for parity runs. */
  int32_t record_parents;   /* SFF* (optimize): 1 = keep the complete parent history (sffgpu_forest_get_parent_history), what
                               an exact tree dump after a given iteration needs (saveIterCheck, src/forest.h:570-578) */
} sffgpu_forest_cfg;

typedef struct {
  int32_t iterations, solved, n_nodes, n_trees, frontier_size, closed_size, n_connected, n_borders;
  uint64_t collide_calls;    /* Environment::Collide calls the reference would have made */
  uint64_t path_free_calls;  /* isPathFree calls the reference would have made */
  uint64_t nn_queries;       /* FLANN queries the reference would have made */
  uint64_t waves;
  uint64_t poses_executed;   /* pose checks launched on the GPU (incl. speculative) */
  uint64_t segments_executed;/* edge checks launched on the GPU (incl. speculative) */
  uint64_t samples_executed; /* edge samples those segments cover */
  uint64_t sweeps;           /* neighbour-sweep launches */
  uint64_t sweep_nodes;      /* sum over sweeps of nodes streamed */
  uint64_t sweep_queries;    /* sum over sweeps of queries */
  uint64_t slow_path_samples;/* samples whose device lists overflowed and were redone on the host path */
  uint64_t grid_rebuilds;    /* times the neighbour grid was re-celled because its overflow list filled up */
  double sweep_ms;           /* device time of the neighbour-query kernels: HIP events on every 8th round
                                (SFFGPU_TIMER_STRIDE), scaled to all rounds - likewise the next two */
  double collide_ms;         /* device time of the pose + segment kernels */
  double sample_ms;          /* device time of the sample+steer kernel */
  double host_ms;            /* host time inside run() not waiting on the device */
  double total_ms;           /* wall time inside run() */
  double query_clock_ms;     /* device-resident engine: neighbour-query kernel time of ALL rounds, bracketed on the device
                                (first wavefront in .. last wavefront out, wall_clock64) - what rocprofv3 reports */
  uint64_t query_clock_launches;
  uint64_t mate_overflow_requeries; /* SFF* host engine: k-nearest queries asked again with a round-sized mate list */
  uint64_t star_rounds;      /* SFF* device engine: rounds whose choose-parent / rewire step ran on the device ... */
  uint64_t star_passes;      /* ... fixed-point passes those rounds took in all (>= 1 each) ... */
  uint64_t star_members;     /* ... k-nearest members (choose-parent / rewire candidates) they looked at */
  uint64_t star_rewires;     /* ... and rewires they applied (src/forest.h:336-348) */
  uint64_t host_fallback_waves; /* device engine: waves finished on the host-replay engine after a device list overflowed */
  double commit_ms;          /* device time of the in-order commit of all rounds (k_commit, the SFF* stage,
                                k_append): the part every rank of a sharded forest repeats; HIP events on the eagerly
                                launched waves, scaled like sweep_ms */
  double exchange_ms;        /* sharded forests: pack + all-gather + unpack of the answer records of all rounds */
  uint64_t graph_launches;   /* device engine: waves launched as one hipGraph replay */
  uint64_t spec_steps;       /* waves of one slot, speculated (k_spec_waves): publish -> evaluate -> commit steps ... */
  uint64_t spec_evaluated;   /* ... attempts its workers evaluated (speculation included) ... */
  uint64_t spec_committed;   /* ... and attempts that were committed (= iterations run by that kernel) */
} sffgpu_forest_stats;

int sffgpu_forest_create(sffgpu_ctx* ctx, const sffgpu_forest_cfg* cfg, const double* roots6, int n_roots,
                         sffgpu_forest** out);
void sffgpu_forest_destroy(sffgpu_forest* f);
/* run to termination, or for at most max_waves waves when max_waves > 0 */
int sffgpu_forest_run(sffgpu_forest* f, int max_waves);
int sffgpu_forest_get_stats(sffgpu_forest* f, sffgpu_forest_stats* out);
/* nodes in global creation order (Solver::allNodes): any output may be NULL */
int sffgpu_forest_get_nodes(sffgpu_forest* f, double* pos6, int32_t* parent, int32_t* tree, int32_t* iter,
                            double* cost, double* dist_parent);
/* SpaceForest::borders entries; returns the count (may exceed cap) or a negative error */
int sffgpu_forest_get_borders(sffgpu_forest* f, int32_t* tree_a, int32_t* tree_b, int32_t* node1, int32_t* node2,
                              double* dist, int cap);
uint64_t sffgpu_forest_fingerprint(sffgpu_forest* f);
/* SFF* with sffgpu_forest_cfg::record_parents = 1: one entry per node creation (roots: parent -1, iteration 0) and per
 * applied rewire (src/forest.h:336-348) - the node, its parent from that iteration on, the iteration - sorted by
 * iteration.  The forest after iteration k consists of the nodes created up to k, each with the parent of its last entry
 * with iteration <= k: what the reference's per-iteration tree dumps show (src/problemStruct.h:255-261).  Returns the
 * number of entries (may exceed cap) or a negative error. */
int sffgpu_forest_get_parent_history(sffgpu_forest* f, int32_t* node, int32_t* parent, int32_t* iteration, int cap);
/* open (frontier) nodes in the order SpaceForest::saveFrontiers writes them (src/forest.h:513-568): the frontier
 * deque, or with priorityBias the first heap of every tree in heap order; returns the count (may exceed cap) */
int sffgpu_forest_get_frontier(sffgpu_forest* f, int32_t* node_ids, int cap);
/* Post-loop path extraction: SpaceForest::getPaths (src/forest.h:420-462) + Solver::getAllPaths
 * (src/problemStruct.h:184-253).  dist = n_trees x n_trees matrix of root-to-root path costs
 * (DBL_MAX where there is none) = Solver::neighboringMatrix; connected = Solver::connectedTrees
 * (tree ids); returns their count.  path_plan copies the node ids of one pair's path (lower node
 * id first, like DistanceHolder) and returns its length. */
int sffgpu_forest_paths(sffgpu_forest* f, double* dist, int32_t* connected, int cap_connected);
int sffgpu_forest_path_plan(sffgpu_forest* f, int i, int j, int32_t* node_ids, int cap);
/* SpaceForest::smoothPaths (src/forest.h:464-511) on the paths extracted by sffgpu_forest_paths: shortcuts
 * every path with batched isPathFree checks; dist receives the updated cost matrix, path_plan the new plans. */
int sffgpu_forest_smooth_paths(sffgpu_forest* f, double* dist);

/* ---------------------------------------------------------------- RRT / RRT* / Multi-T-RRT
 * RapidExpTree<T,R> (src/rrt.h:25-44): constructor :47-83, Solve() :86-99, expandNode :128-322
 * (nearest + steer, RRT* choose-parent / rewire, connect-and-merge of trees). */
typedef struct {
  int32_t dim, optimize, has_goal;
  double goal[6];
  double limits[6];
  double dist_tree, sampling_dist;  /* Problem::distTree, Node::SamplingDistance (scaled) */
  double priority_bias;             /* Problem::priorityBias: probability of steering at the goal */
  int32_t max_iterations;
  uint64_t seed;
  int32_t wave;                     /* iterations speculated per GPU wave: 1 = one by one, 0 = automatic (~sqrt of the tree size) */
  /* LazyTSP<T,R>::runRRT (src/lazy.h:160-284), the Lazy solver's inner planner: ONE tree grown from roots6[0] towards
   * `goal` (has_goal = 0, priority_bias = 0): no tree pick in the RNG stream (:181), k = 2e*log10(tree size + 1)
   * (:199), solved as soon as a new node lies within dist_tree of the goal - no edge check (:258-273).  rng_skip
   * engine words are discarded after seeding: the reference draws all edges' samples from one RandGen. */
  int32_t lazy_edge;
  uint64_t rng_skip;
} sffgpu_rrt_cfg;

typedef struct {
  int32_t iterations, solved, n_nodes, n_live_trees, merges, n_links;
  uint64_t collide_calls, path_free_calls, nn_queries;  /* what the reference would have executed */
  double total_ms;
  uint64_t waves, speculated, committed;                /* wave engine: launched waves, iterations evaluated / kept */
  uint64_t rng_draws;                                   /* engine words consumed so far (rng_skip included) */
  double lazy_distance;                                 /* lazy_edge: edge->distance (src/lazy.h:262; DBL_MAX unsolved, :280) */
} sffgpu_rrt_stats;

typedef struct sffgpu_rrt sffgpu_rrt;
int sffgpu_rrt_create(sffgpu_ctx* ctx, const sffgpu_rrt_cfg* cfg, const double* roots6, int n_roots, sffgpu_rrt** out);
void sffgpu_rrt_destroy(sffgpu_rrt* r);
int sffgpu_rrt_run(sffgpu_rrt* r, int max_iterations); /* 0 = until solved / Problem::maxIterations */
int sffgpu_rrt_get_stats(sffgpu_rrt* r, sffgpu_rrt_stats* out);
/* nodes in creation order; tree = the (possibly merged) tree currently holding the node, root_tree = Node::Root */
int sffgpu_rrt_get_nodes(sffgpu_rrt* r, double* pos6, int32_t* parent, int32_t* tree, int32_t* root_tree,
                         int32_t* iter, double* cost, double* dist_parent);
/* Tree::links entries (src/rrt.h:233); returns the count (may exceed cap) */
int sffgpu_rrt_get_links(sffgpu_rrt* r, int32_t* tree, int32_t* node1, int32_t* node2, double* dist, int cap);
/* RapidExpTree::getConnectedTrees + getPaths (src/rrt.h:324-352, :381-393): the links of the tree that ate the most
 * others become root-to-root paths.  dist = n_trees x n_trees cost matrix (DBL_MAX = none) indexed by Node::Root ids,
 * connected = the central tree's eaten trees + itself; returns their count.  path_plan copies one pair's node ids. */
int sffgpu_rrt_paths(sffgpu_rrt* r, double* dist, int32_t* connected, int cap_connected);
int sffgpu_rrt_path_plan(sffgpu_rrt* r, int i, int j, int32_t* node_ids, int cap);
/* RapidExpTree::smoothPaths (src/rrt.h:354-379), after sffgpu_rrt_paths: shortcuts the plans stored in the central
 * tree's links (isPathFree batched on the GPU).  The reference's neighboringMatrix - what its writers read - holds
 * copies made before (src/rrt.h:350), so costs and sffgpu_rrt_path_plan do not change; the shortened plans are
 * read with sffgpu_rrt_link_plan(k).  Returns the number of link plans. */
int sffgpu_rrt_smooth_paths(sffgpu_rrt* r);
int sffgpu_rrt_link_plan(sffgpu_rrt* r, int k, int32_t* node_ids, int cap);
/* lazy_edge: the solved edge's plan without its goal entry (src/lazy.h:265-272): root ... the node that reached the
 * goal; returns its length (0 = unsolved; may exceed cap) */
int sffgpu_rrt_lazy_plan(sffgpu_rrt* r, int32_t* node_ids, int cap);

/* Multi-GPU wave protocol (one process per GPU; the exchange itself is the caller's RCCL / gloo
 * all-gather).  Every rank holds a full replica of the forest and of the node store; a round is
 *   begin  -> the active slots are drawn and sampled on EVERY rank (replicated, deterministic);
 *             the neighbour sweep, the classification and all collision checks run only for the
 *             candidates this rank owns (candidate i -> rank i % world); their answers are
 *             serialised into an int32 record stream.  *n_words = its length, *done = 1 when the
 *             solver has terminated (then there is nothing to exchange or commit).
 *   records-> copies this rank's stream (n_words int32).
 *   commit -> takes the streams of ALL ranks concatenated in rank order (+ their lengths) and
 *             replays the reference's accept / reject logic in slot order; every rank ends the
 *             round in the same state.  No second collective is needed.
 * sffgpu_forest_run() is this protocol with world == 1. */
int sffgpu_forest_in_wave(sffgpu_forest* f); /* 1 while a wave is open (between its first begin and last commit) */
int sffgpu_forest_round_begin(sffgpu_forest* f, int32_t* n_words, int32_t* done);
int sffgpu_forest_round_records(sffgpu_forest* f, int32_t* words, int cap_words);
int sffgpu_forest_round_commit(sffgpu_forest* f, const int32_t* all_words, int total_words,
                               const int32_t* words_per_rank, int world); /* total_words = length of all_words */


/* The same protocol on the device-resident engine (plain SFF): nothing of a round crosses PCIe.  Every rank runs
 * the replicated kernels (frontier picks, sampling, the in-order commit) itself and evaluates only the candidates it
 * owns; their fixed-size answer records are packed into `send_dev` (sffgpu_forest_exchange_bytes() bytes of DEVICE
 * memory), the caller all-gathers them over RCCL into `recv_dev` (world x that many bytes, rank order) on the stream
 * handed over with sffgpu_ctx_set_stream(), and the commit reads the other ranks' records from there:
 *   sffgpu_forest_dev_wave_begin -> done != 0: the solver has terminated
 *   sffgpu_forest_rounds_per_wave() x { sffgpu_forest_dev_round_eval(send_dev); <all-gather>; sffgpu_forest_dev_round_commit(recv_dev) }
 *   sffgpu_forest_dev_wave_end   -> the one host synchronisation of the wave; fault != 0: a bounded device list
 *                                   overflowed in this wave - the forest is back on the host engine mid-wave and the
 *                                   caller finishes the wave with sffgpu_forest_round_begin / _commit (every rank
 *                                   takes the same decision: the replicas are identical).
 * With world == 1 the buffers may be NULL.  sffgpu_forest_device_engine() tells whether this forest runs on the
 * device engine (plain SFF, wave >= 256, not in libm_sampling mode; SFFGPU_ENGINE=host|device overrides). */
int sffgpu_ctx_set_stream(sffgpu_ctx* ctx, void* hip_stream);   /* NULL: back to the context's own stream */
/* The same exchange driven by the library itself: an RCCL communicator of its own (one rank per process and GPU).
 * Rank 0 makes the 128-byte id (sffgpu_rccl_unique_id), the caller ships it to every rank (any channel), every rank
 * calls sffgpu_ctx_rccl_init(ctx, id, rank, world).  sffgpu_forest_run() of a device-engine forest created with the
 * same rank / world then runs whole waves - replicated kernels, ncclAllGather of the answer records between device
 * buffers on the library's stream, one wave enqueued ahead - with no per-round call from the host language.  Returns
 * SFFGPU_NEED_HOST_EXCHANGE when a bounded device list overflowed: the forest is on the host engine mid-wave and the
 * caller finishes that wave with sffgpu_forest_round_begin / _commit (then calls run again). */
#define SFFGPU_NEED_HOST_EXCHANGE 100
int sffgpu_rccl_unique_id(uint8_t id128[128]);
int sffgpu_ctx_rccl_init(sffgpu_ctx* ctx, const uint8_t id128[128], int rank, int world);
/* The library-driven exchange over a collective of the CALLER's instead of RCCL (process groups that are not RCCL, and
 * the way the library-driven path is tested with several ranks on one GPU): once set, sffgpu_forest_run() of a
 * device-engine forest created with this rank / world calls fn(user, send_dev, recv_dev, words, hip_stream) wherever it
 * would enqueue ncclAllGather - `words` int32 of this rank at send_dev, world x words at recv_dev in rank order, both in
 * device memory.  fn either enqueues the collective on hip_stream or completes it before it returns (it may synchronise
 * the stream); non-zero = failure (the run returns SFFGPU_ERR_HIP).  fn == NULL removes it.  Every rank must make the
 * same sequence of calls: the library enqueues whole waves, one wave ahead, identically on every (identical) replica. */
typedef int (*sffgpu_allgather_fn)(void* user, const void* send_dev, void* recv_dev, size_t words_i32, void* hip_stream);
int sffgpu_ctx_set_allgather(sffgpu_ctx* ctx, sffgpu_allgather_fn fn, void* user, int rank, int world);
int sffgpu_forest_device_engine(sffgpu_forest* f);
long long sffgpu_forest_exchange_bytes(sffgpu_forest* f);
int sffgpu_forest_rounds_per_wave(sffgpu_forest* f);
int sffgpu_forest_dev_wave_begin(sffgpu_forest* f, int32_t* done);
int sffgpu_forest_dev_round_eval(sffgpu_forest* f, void* send_dev);
int sffgpu_forest_dev_round_commit(sffgpu_forest* f, const void* recv_dev);
int sffgpu_forest_dev_wave_end(sffgpu_forest* f, int32_t* fault);

#ifdef __cplusplus
}
#endif
#endif
