"""PPO-side settings (reference: ppo_policy/config.py:11-58)."""
import os

import torch

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")

datapath = {
    "path_data_root": "./dataset",
    "path_init_data": os.path.join("./dataset", "worded_data.pickle"),
    "path_dictionary": os.path.join("./dataset", "dictionary.pickle"),
    "path_train_data": os.path.join("./dataset", "our_dataset.pickle"),
}
Load_Pretrain = True
MaxSeqLen = 1200
TOKEN_COUNT = 150
Pretrain_CKPT = "./ckpt/pretrain_actor.pth"
Output_File_Path = "./gen_midi/pretrain_actor.mid"

ActorConfig = {"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8}
CriticConfig = {"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8}
DiscriConfig = {"MAX_SEQ": 2048, "D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8}
