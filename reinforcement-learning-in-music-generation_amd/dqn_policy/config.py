"""Model dimensions of the DQN-side agent and discriminator (reference: dqn_policy/config.py:11-24)."""
AgentConfig = {"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8}
DiscriConfig = {"D_MODEL": 512, "N_LAYER": 6, "N_HEAD": 8}
