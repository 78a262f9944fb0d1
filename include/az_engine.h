/*
 * az_engine.h -- C-ABI of the MI355X-native batched self-play engine.
 *
 * Drop-in boundary for the hot path of VojtaHavlicek/AlphaZero-Piskvorky
 * (alphazero/mcts.py + alphazero/self_play.py + alphazero/net.py, with games.py,
 * controller.make_policy_value_fn and evaluator.py).  The reference has no FFI of its
 * own: its seams are Python call signatures.  Each entry point below
 * names the reference interface it stands behind; the Python shims in
 * alphazero-piskvorky_amd/ keep those signatures and call this library via ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions: every call returns 0 on success or a negative az_status; nothing throws
 * or aborts across the boundary; the caller owns all input buffers; outputs are written
 * into caller-provided buffers; one engine per GPU; calls on one engine are serialised by
 * the caller (the engine runs its own host threads inside a call, one per lane, and joins them
 * before returning); engines on different GPUs are independent; the calling thread's current
 * HIP device is left as it was.  Pointers suffixed _dev are
 * device pointers (e.g. torch tensor.data_ptr()), all others are host pointers.
 *
 * The library contains NO CPU implementation of the path: az_create fails with
 * AZ_ERR_NO_DEVICE when no gfx950 device is present.
 */
#ifndef AZ_ENGINE_H
#define AZ_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct az_engine az_engine;

typedef enum {
    AZ_OK = 0,
    AZ_ERR_INVALID = -1,     /* bad argument / unsupported configuration            */
    AZ_ERR_NO_DEVICE = -2,   /* no HIP device (the engine has no CPU fallback)      */
    AZ_ERR_HIP = -3,         /* a HIP runtime call failed; see az_last_error        */
    AZ_ERR_ILLEGAL_MOVE = -4,/* games.py:76-77 ValueError("Invalid move")           */
    AZ_ERR_NO_WEIGHTS = -5,  /* net evaluator selected but slot not loaded          */
    AZ_ERR_STATE = -6        /* call sequence error (e.g. export before self-play)  */
} az_status;

enum { AZ_EVAL_NET = 0, AZ_EVAL_SYNTHETIC = 1 };   /* synthetic = deterministic hash evaluator (test hook, mcts.py:87-93 seam) */
enum { AZ_RES_NONE = 0, AZ_RES_X = 1, AZ_RES_O = 2, AZ_RES_DRAW = 3 }; /* constants.py:11-13 'X','O','D' */
enum { AZ_AUG_REFERENCE4 = 4, AZ_AUG_DIHEDRAL8 = 8, AZ_AUG_NONE = 1 };
enum { AZ_TRUNK_F32 = 0, AZ_TRUNK_BF16X3 = 1, AZ_TRUNK_F16X2 = 2 };   /* arithmetic of the conv trunk, az_set_trunk_mode */
enum { AZ_MODEL_PLAIN = 0, AZ_MODEL_RESNET = 1 };  /* net.py GomokuNet | ResidualBlock variant (README.md:72) */

/* Hyper-parameters the reference keeps in constants.py / MCTS.__init__ (mcts.py:87-97). */
typedef struct {
    int32_t board_size;        /* constants.py:2  BOARD_SIZE   (3 .. 15)                         */
    int32_t win_length;        /* constants.py:3  WIN_LENGTH                                     */
    int32_t num_simulations;   /* mcts.py:89      num_simulations (<= 1024)                      */
    int32_t slots;             /* concurrent games resident on this GPU                          */
    double c_puct;             /* mcts.py:90                                                      */
    double dirichlet_alpha;    /* mcts.py:91 (0.3)                                                */
    double dirichlet_weight;   /* mcts.py:92 (0.25)                                               */
    int32_t eval_kind;         /* AZ_EVAL_NET | AZ_EVAL_SYNTHETIC                                 */
    int32_t device;            /* HIP device ordinal                                              */
    /* Optional tables so that the host layer can hand over the reference's own numpy values:
       log_table[N] = float32 log(N + 1e-8) for N = 0..num_simulations (mcts.py:161);
       NULL -> computed with libm logf.                                                          */
    const float *log_table;
    int32_t model;             /* AZ_MODEL_PLAIN (net.py:16-72) | AZ_MODEL_RESNET (config 5)                         */
    /* Lanes inside the engine: the slots are split over `engines` HIP streams, each driven by its own host thread inside
       the library, so that one lane's latency-bound tree / FC kernels run underneath another lane's conv trunk.  All lanes
       take their games from one shared queue (a freed slot of any lane gets the next waiting game) and write into the
       same per-game records, so the episode is the same whatever the number of lanes.  0 = the library chooses
       (1 for boards up to 5x5, else one lane per 128 slots up to 4).  Replaces the worker processes of
       self_play.py:29-45,122-136 (NUM_WORKERS, constants.py:6). */
    int32_t engines;
} az_config;

/* ---- lifecycle ---- */
int az_create(const az_config *cfg, az_engine **out);
void az_destroy(az_engine *e);
const char *az_last_error(const az_engine *e);   /* valid until the next call on e; e may be NULL */

/* ---- weights: replaces state_dict pickling into workers (self_play.py:121,127,36) ----
 * tensors[16] in net.py:37-53 state_dict order (conv1.weight, conv1.bias, conv2.*, conv3.*,
 * policy_conv.*, policy_fc.*, value_conv.*, value_fc1.*, value_fc2.*), fp32, torch layouts.
 * slot 0 = self-play / candidate, slot 1 = arena baseline (evaluator.py:53-62). */
int az_load_weights(az_engine *e, int slot, const float *const *tensors);

/* ResidualBlock variant (engines created with model = AZ_MODEL_RESNET).  The reference no longer ships a forward
 * for it; the topology is fixed by its historical checkpoints (alphazero/models/old/model_20250728_*.pt) and the
 * block by legacy/resnet/example.py:9-27: conv(4->64)+BN+ReLU, 3 x {conv+BN+ReLU, conv+BN, +skip, ReLU},
 * policy_conv(64->2)+BN+ReLU -> policy_fc(2n^2 -> n^2), value_conv(64->1)+BN+ReLU -> value_fc1(n^2 -> 64) -> ReLU ->
 * value_fc2 -> tanh.  tensors[24], fp32, eval-mode BatchNorm already FOLDED into conv weight/bias by the caller:
 *   0,1 stem w[64,4,3,3], b[64];  2+2i, 3+2i (i = 0..5) res{1,2,3}.conv{1,2} w[64,64,3,3], b[64];
 *   14,15 policy_conv w[2,64], b[2];  16,17 value_conv w[1,64], b[1];  18,19 policy_fc w[n^2,2n^2], b[n^2];
 *   20,21 value_fc1 w[64,n^2], b[64];  22,23 value_fc2 w[64], b[1]. */
int az_load_weights_resnet(az_engine *e, int slot, const float *const *tensors);

/* ---- batched net evaluation: controller.make_policy_value_fn (controller.py:33-55) ----
 * boards[count][n*n] absolute cells (0 empty, 1 X, 2 O), players[count] side to move,
 * lasts[count] last action (r*n+c, -1 none).  Outputs: logits/policy [count][n*n], value[count].
 * Any output pointer may be NULL. */
int az_net_eval(az_engine *e, int slot, int count, const uint8_t *boards, const uint8_t *players,
                const int16_t *lasts, float *logits, float *policy, float *value);

/* ---- single-position search: MCTS.run (mcts.py:101-183) ----
 * noise: Dirichlet sample over the legal cells in row-major order, or NULL for add_root_noise=False;
 * u: the uniform np.random.choice draws (mcts.py:177).  temperature as float64 (np.float64 schedules).
 * Outputs (any may be NULL): pi[n*n] float32, action, visits[n*n], W[n*n] float64, prior[n*n]. */
int az_search(az_engine *e, int slot, const uint8_t *board, int player, int last, double temperature,
              const double *noise, double u, float *pi, int32_t *action, int32_t *visits, double *W,
              float *prior);

/* ---- the same search with the evaluator outside the engine: the policy_value_fn plugin seam (mcts.py:87-93) ----
 * The reference's MCTS takes ANY callable state -> (policy float32[n,n], value float) (controller.py:39-53 is just the
 * one the training loop uses).  az_search_callback keeps that seam: select / expand / backup / pi extraction run on the
 * GPU as in az_search, and each of the num_simulations + 1 evaluations (mcts.py:109,137) is handed to `fn` on the host:
 * board[n*n] absolute cells (0 empty, 1 X, 2 O), player = side to move there, last = last action (-1 none); fn writes
 * policy[n*n] (used as priors exactly as given: mcts.py:63 float(policy[r, c]), no masking, no renormalisation) and
 * *value (from the point of view of `player`), and returns 0 (anything else aborts the search with AZ_ERR_INVALID).
 * One host round trip per simulation: a compatibility path, not a fast one.  Needs no weights; not combinable with
 * virtual-loss batching or subtree reuse.  Other arguments and outputs as az_search. */
typedef int (*az_eval_callback)(void *user, const uint8_t *board, int player, int last, float *policy, float *value);
int az_search_callback(az_engine *e, const uint8_t *board, int player, int last, double temperature, const double *noise,
                       double u, az_eval_callback fn, void *user, float *pi, int32_t *action, int32_t *visits, double *W,
                       float *prior);

/* ---- self-play episode: SelfPlayManager.generate_self_play + _worker (self_play.py:29-77,110-159) ----
 * Plays games with ids [0, num_games); game g draws its randomness from numpy-compatible
 * RandomState(seed0 + g) (dirichlet then one uniform per ply) unless a tape is given.
 * temperature_table[m], m = 0..n*n : temperature_schedule(m) evaluated by the host layer
 * (self_play.py:24-26,53); NULL -> the reference default (exp(-m/100)+0.01)/1.01 via libm.
 * max_plies > 0 cuts every game after that many plies (fixtures / benchmarking), 0 = play to the end.
 * Results stay resident on the device until the next az_selfplay / az_destroy. */
typedef struct {
    uint64_t seed0;
    int32_t num_games;
    int32_t max_plies;
    const double *temperature_table;
    /* optional explicit tapes (tests): noise_tape[g] concatenated per game with stride tape_stride doubles,
       u_tape[g][n*n]; NULL -> generated from seed0 + g */
    const double *noise_tape;
    const double *u_tape;
    int64_t tape_stride;
} az_selfplay_args;

typedef struct {
    int64_t games, plies, records;      /* records = plies (one per position, before augmentation)      */
    int64_t simulations;                /* MCTS simulations run (mcts.py:123)                            */
    int64_t expansions;                 /* non-terminal leaf evaluations (mcts.py:136-138)               */
    int64_t root_evals;                 /* root evaluations (mcts.py:109), one per ply                   */
    int64_t terminal_hits;              /* simulations ending in a terminal leaf (mcts.py:132-134)       */
    int64_t depth_sum;                  /* sum of selection depths (mean depth = depth_sum/simulations)  */
    int64_t steps;                      /* lock-step evaluation batches launched                         */
    double seconds;                     /* device wall time of the episode                               */
    double nn_seconds;                  /* HIP-event time inside the net kernels                         */
    double trunk_seconds;               /* ... of which the fused conv trunk (dominant kernel)           */
    int64_t trunk_launches;
    int64_t trunk_boards;               /* boards evaluated by the trunk kernel (all launches)           */
    double step_seconds;                /* HIP-event time inside the tree kernel k_step (select/expand/backup) */
    int64_t duplicate_leaves;           /* virtual-loss batching: simulations that met a leaf already pending in their batch */
    int64_t cache_lookups, cache_hits;  /* evaluation cache: positions looked up / found (a hit skips the net kernels)  */
    /* Host side of the RNG tapes (mcts.py:114,177 draws, produced on the host ahead of the games): the longest time any
       lane's driver thread was blocked because the tape wave of its next ply was not yet produced or uploaded -- 0 when the
       producer keeps ahead of the games; a tape-bound run (many ranks on few cores) shows up here instead of looking like
       a slow GPU.  tape_threads = producer threads of this engine, host_cpus = CPUs the process may use (affinity mask
       and cgroup quota) that the count was sized from. */
    double tape_wait_seconds;
    int64_t tape_threads, host_cpus;
} az_counters;

int az_selfplay(az_engine *e, const az_selfplay_args *args, az_counters *out);

/* The same episode in pieces (benchmarks, pipelined callers): begin uploads tapes and fills the slots;
 * step plays up to max_steps plies of every active game in lock step (one step = MCTS.run for each
 * active game = 1 + num_simulations evaluation batches) and reports the number of still-active slots and,
 * optionally, the counters so far; end finalises the per-game results. */
int az_selfplay_begin(az_engine *e, const az_selfplay_args *args);
int az_selfplay_step(az_engine *e, int max_steps, int32_t *active_out, az_counters *progress);
int az_selfplay_end(az_engine *e, az_counters *out);

/* Per-game summary of the last episode: nply[g], result[g] (AZ_RES_*). */
int az_selfplay_games(az_engine *e, int32_t *nply, int32_t *result);

/* Raw per-ply records of the last episode, game-major then ply (tests/diagnostics):
 * boards[records][n*n] absolute cells before the move, movers, lasts, actions, pis[records][n*n],
 * visits[records][n*n], z[records] (self_play.py:71; 99 when the game was cut by max_plies). */
int az_selfplay_records(az_engine *e, uint8_t *boards, uint8_t *movers, int16_t *lasts, int16_t *actions,
                        float *pis, int32_t *visits, int8_t *z);

/* Packed records for the episode-end exchange (RCCL gather over xGMI):
 * az_record_bytes() per record: mover-relative bit-planes, last move, pi float32[n*n], z.
 * az_selfplay_pack writes records*az_record_bytes bytes to a DEVICE buffer. */
int64_t az_record_bytes(const az_engine *e);
int az_selfplay_pack(az_engine *e, void *packed_dev);

/* ---- multi-GPU: the episode-end exchange inside the library (replaces result_queue.put / get across workers,
 * self_play.py:73,140, and the tally of evaluator.py:106-109 across ranks) ----
 * One engine per rank = per GPU; ranks play disjoint game ids (the caller passes seed0 + first id and its share of the
 * games to az_selfplay / az_arena), so the only communication is at episode end.  The library calls RCCL directly on the
 * engine's own stream (librccl.so.1 is bound at run time: the copy a host framework has already loaded, else the
 * system's); nothing here needs torch.distributed.
 *   az_dist_unique_id   one rank creates the communicator id (ncclGetUniqueId) and hands its AZ_DIST_ID_BYTES bytes to the
 *                       other ranks over whatever channel the caller has (a file, MPI, a launcher's store);
 *   az_dist_init        collective: every rank joins with the same id (ncclCommInitRank on the engine's device);
 *   az_dist_counts      collective: counts[world] = records of every rank's last episode (ncclAllGather of one int64);
 *   az_dist_gather_records  collective: the packed records (az_record_bytes each, as az_selfplay_pack writes them) of all
 *                       ranks, in rank order, into packed_dev on rank dst -- grouped ncclSend / ncclRecv of the TRUE sizes, no
 *                       padding to the largest rank, nothing sent to ranks that do not train; dst = -1: every rank
 *                       receives everything.  packed_dev (DEVICE) needs room for sum(counts) records on a receiving rank
 *                       and may be NULL elsewhere.  A rank without an episode contributes 0 records.
 *   az_dist_allreduce_sum   collective: element-wise sum of n int64 values over the ranks (the arena's wins / losses / draws);
 *   az_dist_broadcast   collective: bytes of a DEVICE buffer from rank root to every rank (new weights after train_step).
 * az_dist_rank / az_dist_world: 0 / 1 before az_dist_init.  Errors: AZ_ERR_STATE without az_dist_init, AZ_ERR_HIP with the
 * RCCL message in az_last_error when RCCL cannot be loaded or a call fails. */
#define AZ_DIST_ID_BYTES 128
int az_dist_unique_id(void *id);
int az_dist_init(az_engine *e, const void *id, int rank, int world);
int az_dist_rank(const az_engine *e);
int az_dist_world(const az_engine *e);
int az_dist_counts(az_engine *e, int64_t *counts);
int az_dist_gather_records(az_engine *e, int dst, void *packed_dev);
int az_dist_allreduce_sum(az_engine *e, int64_t *values, int n);
int az_dist_broadcast(az_engine *e, void *buf_dev, int64_t bytes, int root);

/* Training examples: (state f32[4,n,n], pi f32[n,n], z) with the augmentation of
 * SelfPlayManager._augment_symmetries fused into the encode (self_play.py:94-108,146-148):
 * AZ_AUG_REFERENCE4 reproduces the reference (state rot k*90deg, pi rot 90deg once, Q16),
 * AZ_AUG_DIHEDRAL8 is the correct 8-fold group, AZ_AUG_NONE emits each position once.
 * Reads `records` packed records from packed_dev (this rank's or gathered ones) and writes
 * records*aug examples to DEVICE buffers, order (record, k). */
int az_examples_from_packed(az_engine *e, const void *packed_dev, int64_t records, int aug,
                            float *states_dev, float *pis_dev, float *z_dev);

/* Training batch from a device-resident replay ring of packed records (replay_buffer.py:26-39 sample_batch +
 * controller.py:23-31 collate, with no host round trip): example i = symmetry sym_dev[i] (0..7, dihedral group;
 * 0..3 are the rotations) of record idx_dev[i].  reference_pi != 0 rotates pi once whatever the symmetry, like
 * self_play.py:105 does.  All pointers are DEVICE pointers. */
int az_examples_gather(az_engine *e, const void *packed_dev, const int64_t *idx_dev, const int32_t *sym_dev,
                       int count, int reference_pi, float *states_dev, float *pis_dev, float *z_dev);

/* ---- arena: ModelEvaluator.evaluate (evaluator.py:38-122) ----
 * candidate = slot 0 plays X, baseline = slot 1 plays O, odd game index => O moves first;
 * per-game RandomState(seed0 + g), one uniform per ply; temperature_table[step] =
 * evaluator.temperature_schedule(step) (NULL -> 0.3*exp(-step/4) via libm). */
typedef struct {
    uint64_t seed0;
    int32_t num_games;
    const double *temperature_table;
    const double *u_tape;               /* optional [num_games][n*n] */
} az_arena_args;
typedef struct { int32_t wins, losses, draws, total; double win_rate; } az_arena_result;
int az_arena(az_engine *e, const az_arena_args *args, az_arena_result *out, int32_t *results, int16_t *actions,
             int32_t *nply);

/* ---- rules only: Gomoku.apply_action / is_terminal / get_game_result (games.py:64-82,133-179) ----
 * Replays `games` action lists (actions[g][max_len], r*n+c, -1 = end of list) from the empty board with X to move, on the
 * device, with the same bit-plane win test the search kernels use.  Like the reference, apply_action does not look at
 * the terminal flag (a list may continue past the end of the game; the winner stays the first one, games.py:140-141).
 * Outputs (term_before may be NULL): term_before[g][i] = is_terminal() before move i; boards[g][n*n] final cells
 * (0 empty, 1 X, 2 O); players[g] = side to move afterwards; results[g] = AZ_RES_*; first_illegal[g] = index of the
 * first illegal action of the list (occupied cell or out of range: games.py:76-77 ValueError), -1 if none -- the replay
 * of that game stops there. */
int az_rules_replay(az_engine *e, int games, int max_len, const int16_t *actions, uint8_t *term_before, uint8_t *boards,
                    int32_t *players, int32_t *results, int32_t *first_illegal);

/* ---- host-side numpy-compatible RNG (legacy MT19937 RandomState; mcts.py:114,177) ---- */
int az_rng_selfplay_tape(uint64_t seed, int board_size, double alpha, int max_plies, double *noise, double *u);
int az_rng_uniforms(uint64_t seed, int count, double *u);

/* Opt-in search upgrade the reference lists as a TODO (mcts.py:17-22 "subtree reuse", mcts.py:106 builds a new root
 * every run): after a self-play move the chosen child's subtree becomes the next ply's tree.  The retained root keeps
 * its priors and visit statistics, receives the ply's fresh Dirichlet sample with the arithmetic of a new root
 * (mcts.py:113-116), is NOT evaluated again, and the search runs only the simulations that top its children's visits
 * up to num_simulations, so pi is still a distribution over num_simulations visits.  Off by default: with it the visit
 * counts differ from the reference's (parity is then against the oracle's restatement of this rule, "parity
 * unpinned" by the reference).  Self-play only; arena games and az_search always start from a fresh root.  Needs
 * num_simulations <= 1023.  Not allowed while an episode is open. */
int az_set_subtree_reuse(az_engine *e, int on);

/* Opt-in: virtual-loss batching within a search, the third item of the reference's TODO list (mcts.py:17-22); its
 * simulation loop (mcts.py:123-141) is strictly sequential: select one leaf, evaluate it, back it up.  With `leaves` = L > 1
 * the num_simulations simulations of a search run in batches of L: the leaves of a batch are selected one after another
 * from the same tree, each selection counting the earlier ones of its batch as one visit that lost on every edge of their
 * paths (N + 1, W - 1 in mcts.py:73's formula), then the L leaves are evaluated together, and the simulations are
 * finished in selection order exactly like mcts.py:136-141 (the virtual visit is taken back first).  A selection that
 * ends on a leaf an earlier one of its batch is waiting on does not evaluate again: it backs up that leaf's value
 * (az_counters.duplicate_leaves).  Every search still makes exactly num_simulations simulations, in
 * ceil(num_simulations / L) dependent evaluation batches instead of num_simulations -- the lever for the latency-bound
 * uses (az_search, az_arena, episode tails).  Visit counts differ from the reference's sequential search, so parity is
 * against the oracle's restatement of this rule ("parity unpinned" by the reference); L = 1 is the reference's loop.
 * 1 <= leaves <= 32.  Combines with random-symmetry leaf evaluation; not with subtree reuse.  Not allowed while an episode
 * is open. */
int az_set_virtual_loss(az_engine *e, int leaves);

/* Opt-in: evaluation cache, the first item of the reference's TODO list (mcts.py:17 DEFAULT_CACHE_SIZE = 500_000,
 * mcts.py:22 "TODO: add caching").  A table in HBM keyed on what the net sees -- mover planes, opponent planes, last
 * move (games.py:86-129), which weight slot -- holding the net's raw outputs (policy logits, value-head hidden row).
 * A leaf or root whose position is in the table skips the conv trunk and the FC layers; the values are the very floats
 * those kernels would produce again, so visit counts, pi and moves are bit-identical with the cache on or off.
 * Shared by all lanes of the engine and kept across plies, games and episodes; every az_load_weights* call invalidates
 * it.  entries = 0 switches it off (and frees it); otherwise rounded up to a power of two, (96 + roundup(n*n, 64)) * 4
 * bytes each.  Hits are reported in az_counters.cache_lookups / cache_hits.  Not allowed while an episode is open. */
int az_set_eval_cache(az_engine *e, int64_t entries);

/* Opt-in: random-symmetry leaf evaluation, the optional half of SURVEY §8f-2 (the reference's README.md:61,82 names the 8-fold
 * symmetry of the board, its code only uses it to augment the examples and leaves Gomoku.rot90 / flip unused,
 * games.py:183-197).  With it every net evaluation of a search -- the root (mcts.py:109) and each expanded leaf
 * (mcts.py:137) -- shows the net one of the 8 dihedral symmetries of the position and maps the policy back: priors of a
 * board cell = softmax entry of the image cell it was moved to; the value is taken as is.  The symmetry of evaluation idx
 * (0 = root, s + 1 = simulation s) of game g at ply p is a fixed hash of (low 32 bits of seed0 + g, p, idx) -- the game is
 * named by its seed, which a rank playing the id block [lo, hi) of a larger episode passes as seed0 + lo -- so runs are
 * reproducible and independent of slots, lanes and ranks (az_search / az_search_callback: key 0).  Visit counts differ from the reference's (its net always sees the position
 * unrotated); parity is against the oracle's restatement of this rule (orc_cfg.leaf_sym, "parity unpinned" by the
 * reference).  Lock-step pipeline and persistent search kernel alike; combines with virtual-loss batching (simulation s of a batch is evaluation s + 1 like
 * in the sequential loop), with the evaluation cache (the symmetry is part of the key: a hit is the evaluation under that
 * symmetry, records are unchanged) and with subtree reuse (a retained root is not evaluated again; simulation s stays
 * evaluation s + 1); az_search_callback evaluates positions as they are.  Not allowed while an episode is open. */
int az_set_leaf_symmetry(az_engine *e, int on);

/* Opt-in: fp32-emulating conv trunks.  AZ_TRUNK_F32 (default) computes the net's forward (net.py:55-72) on the float32
 * matrix instruction in the build's canonical fp order: bit-identical to the oracle.  The other two run every conv but the
 * first (99 % of the net's arithmetic) and the 1x1 head convs on the 16 x faster 16-bit matrix instructions with split
 * operands and float32 accumulation:
 *   AZ_TRUNK_BF16X3  three bfloat16 parts per operand, the six largest cross products: float32's exponent range, 2.67 x
 *                    ceiling over the float32 instruction;
 *   AZ_TRUNK_F16X2   two float16 parts per operand (the low part scaled by 2^11), three cross products in two accumulators:
 *                    5.3 x ceiling; float16's range -- activations saturate at 65504 and a weight set with |w| >= 65504 is
 *                    refused (AZ_ERR_INVALID from this call or from the next az_load_weights*, which puts the engine back on
 *                    AZ_TRUNK_F32).
 * Both give float32-like accuracy (|logit| 2e-5, |P| 1e-6, |value| 2e-6 against the oracle, the tolerances already granted
 * against the Python reference's torch numbers), NOT bit-identical results, so visit counts can differ from the reference's
 * on near-tied PUCT scores (measured: DESIGN.md section 4).  One kernel for every occupancy (no split / persistent variants),
 * so results do not depend on the slot count.  Both nets.  Invalidates the evaluation cache.  Not allowed while an episode
 * is open. */
int az_set_trunk_mode(az_engine *e, int mode);
int az_get_trunk_mode(const az_engine *e);
/* Host-side helper (needs no GPU, like az_rng_*): the 16-bit parts a weight is split into for `mode` -- bfloat16 hi, mid, lo
 * (x = hi + mid + lo) or float16 hi, lo (x = hi + lo / 2048) as raw bit patterns.  Returns the number of parts written, or
 * AZ_ERR_INVALID for an unknown mode or a value outside float16's range in AZ_TRUNK_F16X2. */
int az_emul_split(int mode, float x, uint16_t *parts);

/* HIP-event timing of every trunk / FC / tree-step launch (az_counters.trunk_seconds, nn_seconds, step_seconds);
 * off by default: four events per evaluation batch cost a few microseconds of stream time, which matters on small boards.
 * While it is on, the lanes of the engine play one after another instead of concurrently, so that every kernel is timed
 * alone on the GPU (roofline calibration). */
int az_set_profiling(az_engine *e, int on);

int az_get_counters(const az_engine *e, az_counters *out);

/* Number of lanes the engine runs (az_config.engines after the library's choice for 0). */
int az_get_lanes(const az_engine *e);

/* Which kernels ran the searches of the last (or the open) episode: 0 = the lock-step pipeline, one conv-trunk, FC and
 * tree launch per evaluation batch; g > 0 = the persistent search kernel (csrc/az_search.h), one launch per ply with g
 * games per workgroup and their trees resident in LDS.  The library picks the persistent kernel by itself whenever it
 * applies -- boards up to 7x7 whose (num_simulations + 1) tree rows fit into LDS, either net or the synthetic evaluator,
 * float32 trunk, no virtual-loss batching (the evaluation cache, subtree reuse and the leaf symmetry it knows) -- because
 * its results are bit-identical; AZ_PERSIST=0 in the environment keeps the lock-step pipeline. */
int az_get_persistent(const az_engine *e);

#ifdef __cplusplus
}
#endif
#endif /* AZ_ENGINE_H */
