"""Drop-in for /root/reference/ppo_policy/inference.py::testing (generation with the PPO actor): recurrent-form
`Actor_Transformer`, `TOKEN_COUNT` tokens drawn attribute-wise from Categorical(softmax(logits)), entirely on the
device (rlmg_amd.generation.categorical_rollout).  The reference converts the tokens with
prepare_data.tuple_events_to_midi (miditoolkit tooling, out of scope); here they are saved as .npy.
"""
import os
import sys

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import data as cwdata, generation  # noqa: E402

try:
    from config import datapath
    from model import Actor_Transformer
except ImportError:
    from .config import datapath
    from .model import Actor_Transformer

TOKEN_COUNT = 1000
Pretrain_CKPT = "./ckpt/actor_pretrain.pt"
Output_File_Path = "./gen_midi/ppo_song.npy"


def testing(token_count=None, carry_memory=False, log=print):
    os.makedirs(os.path.dirname(Output_File_Path) or ".", exist_ok=True)
    dictionary, _ = cwdata.load_ppo(datapath["path_dictionary"], datapath["path_train_data"], n_seq=1, T=64)
    event2word, word2event = dictionary
    num_token = [len(event2word[k]) for k in event2word.keys()]
    log("Num of class of token:", num_token)
    model = Actor_Transformer(num_token, is_training=False).cuda()
    if os.path.exists(Pretrain_CKPT):
        model.load_state_dict(torch.load(Pretrain_CKPT, map_location="cuda"))
    else:
        log("[*] %s not found: sampling from freshly initialised weights" % Pretrain_CKPT)
    model.eval()
    song = generation.categorical_rollout(model, TOKEN_COUNT if token_count is None else token_count,
                                          carry_memory=carry_memory)
    np.save(Output_File_Path, song)
    log("====== Finish ====== ")
    return song


if __name__ == "__main__":
    testing()
