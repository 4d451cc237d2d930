"""oracle/ -- TEST INFRASTRUCTURE, not product code.

CPU restatement (plain PyTorch fp32/fp64 + numpy) of the reference's arithmetic for the hot path
named in BASELINE.json: compound-word Linear Transformer + AIRL / PPO / DQN training.  Every
function cites the reference file:line it follows (paths relative to /root/reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the timed CPU baseline.  The product package
(reinforcement-learning-in-music-generation_amd/) never imports it and raises if libcwlt.so is
missing instead of falling back to anything here.

Pinning status (see DESIGN.md "Oracle"):
  * wrappers (embeddings, in_linear, positional encoding, heads, CE loss, PPO/DQN math): PINNED --
    the reference's own dqn_policy/model.py and ppo_policy/model.py were imported and run in the
    build container; golden vectors are committed under tests/golden/ with the generating script.
  * encoder body (fast_transformers 0.4.0 causal-linear attention / encoder layers): PARITY
    UNPINNED -- the package is a third-party dependency absent from /root/reference and from this
    image and cannot be fetched; oracle/cla.py + oracle/ft_encoder.py restate its published
    algorithm (Katharopoulos et al. 2020, alg. 1) in three mutually-checking forms.
"""
