"""Constants of the rollout path; names and values follow reference parameters.py:1-28."""
C_PUCT = 5          # parameters.py:8
EPS = 0.25          # parameters.py:10  Dirichlet mixing weight
ALPHA = 0.2         # parameters.py:12  Dirichlet concentration
PLAYOUT = 1600      # parameters.py:14  simulations per move (BASELINE configs use 200/400/800)
DATA_DIR = "data"   # parameters.py:16
MODEL_DIR = "models"  # parameters.py:18
BATCH_SIZE = 2048   # parameters.py:20
EPOCHS = 10
KL_TARG = 0.02
CHECK_FREQ = 10
LOG_LEVEL = 1
