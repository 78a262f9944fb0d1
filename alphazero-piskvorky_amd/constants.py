"""Hyper-parameters under the names the reference's constants.py exposes (constants.py:1-35).
Edit here or pass explicit arguments; the engine itself takes an explicit az_config."""
BOARD_SIZE = 5
WIN_LENGTH = 4
NUM_EPISODES = 10
NUM_WORKERS = 6            # unused by the GPU path (one engine per GPU replaces worker processes)
DEVICE = "cuda"
X, O, DRAW = "X", "O", "D"
NUM_SELF_PLAY_GAMES = 100
NUM_SELF_PLAY_SIMULATIONS = 200
SELF_PLAY_EXPLORATION_CONSTANT = 2.0
BUFFER_CAPACITY = 40_000
TEMPERATURE_SCHEDULE_HALFTIME = 100
TEMPERATURE_BASELINE = 0.01
BATCH_SIZE = 1024
BATCHES_PER_EPISODE = 10
NUM_EPOCHS = 3
LEARNING_RATE = 5 * 1e-5
MODEL_DIR = "models"
EVALUATION_GAMES = 51
NUM_EVAL_SIMULATIONS = 200
EVAL_EXPLORATION_CONSTANT = 2.0
EVAL_TEMPERATURE = 0.3
EVAL_TEMPERATURE_SCHEDULE_HALFTIME = 4
# engine knobs (not in the reference)
CONCURRENT_GAMES = 1024    # game slots per GPU
ENGINES_PER_GPU = 4        # slots are split over this many engines/streams (tree + FC kernels overlap the conv trunk)
