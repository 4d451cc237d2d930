"""Drop-in for /root/reference/dqn_policy/agent_pretrain.py: `TransformerModel` (the same network as
model.LinearTransformer), `train()` (MODE='train') and `generate()` (MODE='inference': token-by-token sampling on
the recurrent form, rlmg_amd.generation; songs are written with rlmg_amd.midi.write_midi).
"""
import datetime
import os
import sys
import time

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import data as cwdata, dist as rdist, ops  # noqa: E402

try:
    from model import LinearTransformer, network_paras
    from saving import Saver
except ImportError:
    from .model import LinearTransformer, network_paras
    from .saving import Saver

MODE = "train"
path_data_root = "/data/dataset_Pop1K7/representations/uncond/cp/ailab17k_from-scratch_cp"
path_train_data = os.path.join(path_data_root, "train_data_linear.npz")
path_dictionary = os.path.join(path_data_root, "dictionary.pkl")
D_MODEL, N_LAYER, N_HEAD = 512, 12, 8
path_exp = "exp"
batch_size = 4
gid = 0
init_lr = 0.0001


class TransformerModel(LinearTransformer):
    """agent_pretrain.py:213 re-declares model.LinearTransformer under this name."""


def train(n_epoch=None, compute_dtype=torch.float32, log=print):
    """agent_pretrain.py:485-632: sequential batches of 4, loss = mean of the 6 CE losses, zero_grad / backward /
    clip_grad_norm_(3) / Adam(1e-4); loss-banded checkpoints under ./ckpt."""
    n_epoch = int(os.environ.get("CWLT_N_EPOCH", 4000)) if n_epoch is None else n_epoch
    max_grad_norm = 3
    rank, local, world = rdist.init_from_env()
    torch.cuda.set_device(local)
    dictionary, train_data = cwdata.load_dqn(path_train_data, path_dictionary)
    event2word, word2event = dictionary
    n_class = [len(event2word[k]) for k in event2word.keys() if k != "type"]
    log("num of classes:", n_class)
    net = TransformerModel(n_class)
    net.cuda()
    net.train()
    net.compute_dtype = compute_dtype
    log("n_parameters: {:,}".format(network_paras(net)))
    saver_agent = Saver(path_exp) if rank == 0 else None         # agent_pretrain.py:495-513: exp/log.txt
    if saver_agent:
        saver_agent.add_summary_msg(" > params amount: {:,d}".format(network_paras(net)))
    sync = rdist.GradSync(net.parameters())
    optimizer = ops.graph_adam(net.parameters(), lr=init_lr)     # torch.optim.Adam, single-kernel form on the GPU
    train_x = np.concatenate((train_data["x"][:, :, :3], train_data["x"][:, :, 4:]), axis=2)
    train_y = np.concatenate((train_data["y"][:, :, :3], train_data["y"][:, :, 4:]), axis=2)
    train_mask = train_data["mask"]
    num_batch = len(train_x) // (batch_size * world)
    start_time = time.time()
    epoch_loss = float("nan")
    for epoch in range(n_epoch):
        acc_loss, acc_losses = 0.0, np.zeros(6)
        for bidx in range(num_batch):
            if saver_agent:
                saver_agent.global_step_increment()
            st = batch_size * (bidx * world + rank)              # rank-strided batches under data parallelism
            batch_x = torch.from_numpy(train_x[st:st + batch_size]).long().cuda()
            batch_y = torch.from_numpy(train_y[st:st + batch_size]).long().cuda()
            batch_mask = torch.from_numpy(train_mask[st:st + batch_size]).float().cuda()
            losses = net.train_step(batch_x, batch_y, batch_mask)
            loss = (losses[0] + losses[1] + losses[2] + losses[3] + losses[4] + losses[5]) / 6
            sync.zero_grad()
            loss.backward()
            sync.finish()
            sync.clip_grad_norm_(max_grad_norm)
            optimizer.step()
            acc_losses += np.array([l.item() for l in losses])
            acc_loss += loss.item()
            if saver_agent:
                saver_agent.add_summary("batch loss", loss.item())
        runtime = time.time() - start_time
        epoch_loss = acc_loss / max(1, num_batch)
        if world > 1:
            # every rank must take the same stop / checkpoint-band decision (the reference is one process): use the
            # mean over ranks of the per-rank epoch losses
            t = torch.tensor([epoch_loss], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t)
            epoch_loss = t.item() / world
        if saver_agent:
            saver_agent.add_summary("epoch loss", epoch_loss)
            saver_agent.add_summary("epoch each loss", "{:04f}, {:04f}, {:04f}, {:04f}, {:04f}, {:04f}\r".format(
                *(acc_losses / max(1, num_batch))))
        log("Epoch: {}/{} | Loss: {} | time: {}".format(epoch, n_epoch, epoch_loss,
                                                        str(datetime.timedelta(seconds=runtime))))
        if epoch_loss <= 0.05:                     # agent_pretrain.py:611-613 -- on EVERY rank (same epoch_loss)
            log("Finished")
            return epoch_loss
        if rank == 0:
            os.makedirs("./ckpt", exist_ok=True)
            if 0.4 < epoch_loss <= 0.8:
                name = "trainloss_" + str(int(epoch_loss * 10) * 10) + ".pt"
            elif 0.05 < epoch_loss <= 0.40:
                name = "trainloss_" + str(int(epoch_loss * 100)) + ".pt"
            else:
                name = "trainloss_" + str(int(epoch_loss * 100)) + "_high.pt"
            torch.save({"epoch": n_epoch, "model_state_dict": net.state_dict(),
                        "optimizer_state_dict": optimizer.state_dict()}, os.path.join("./ckpt", name))
    return epoch_loss


path_gendir = "gen_midis"
num_songs = 5
bar_production = 50                               # testing-no-type-cp.py:35


def generate(n_songs=None, bar_cond=None, max_tokens=None, log=print, device_sampling=False):
    """agent_pretrain.py:636-706 / testing-no-type-cp.py:182-260: build the recurrent-form net, load
    ./ckpt/_params.pt when present, sample `num_songs` songs, write get_<i>.mid + runtime_stats.json."""
    from rlmg_amd import generation, midi
    dictionary, _ = cwdata.load_dqn(path_train_data, path_dictionary, n_seq=1, T=64)
    event2word, word2event = ({k: v for k, v in d.items() if k != "type"} for d in dictionary)
    n_class = [len(event2word[k]) for k in event2word.keys()]
    net = TransformerModel(n_class, is_training=False)
    net.cuda()
    net.eval()
    path_saved_ckpt = os.path.join("./ckpt/" + "_params.pt")
    if os.path.exists(path_saved_ckpt):
        log("[*] load model from:", path_saved_ckpt)
        sd = torch.load(path_saved_ckpt)
        net.load_state_dict(sd.get("model_state_dict", sd))
    else:
        log("[*] %s not found: sampling from freshly initialised weights" % path_saved_ckpt)
    return generation.generate(net, word2event, n_songs=num_songs if n_songs is None else n_songs,
                               bar_cond=bar_production if bar_cond is None else bar_cond, path_gendir=path_gendir,
                               write_midi=midi.write_midi, max_tokens=max_tokens, log=log,
                               device_sampling=device_sampling)


if __name__ == "__main__":
    if MODE == "train":
        train()
    elif MODE == "inference":
        generate()
