"""Dataset plumbing for the entry scripts: the reference's on-disk formats when the files exist, synthetic
data of the same schema otherwise (the dataset and the pretrained weights are external downloads,
README.md:23,27 -- unavailable offline).

  dqn: train_data_linear.npz {x, y (N, 3584, 7) int, mask (N, 3584)} + dictionary.pkl = (event2word, word2event)
       keyed tempo, chord, bar-beat, type, pitch, duration, velocity      (IRL_dqn_train.py:389-391,418-420)
  ppo: dictionary.pickle + our_dataset.pickle {train_x, train_y (N, 1200, 6), mask (N, 1200)}
       (ppo_train.py:432-449, preprocess.py:66-72)
"""
import os
import pickle

import numpy as np

DQN_KEYS = ("tempo", "chord", "bar-beat", "type", "pitch", "duration", "velocity")
DQN_N = (56, 135, 18, 3, 87, 18, 25)             # IRL_dqn_train.py:403 (+ 3 `type` classes)
PPO_KEYS = ("Tempo", "Bar", "Position", "Pitch", "Duration", "Velocity")
PPO_N = (49, 19, 19, 89, 67, 25)                 # prepare_data.py:247-291


def _synth(n_seq, T, n_per_field, seed):
    rng = np.random.default_rng(seed)
    x = np.stack([rng.integers(0, n, (n_seq, T)) for n in n_per_field], -1).astype(np.int64)
    y = np.stack([rng.integers(0, n, (n_seq, T)) for n in n_per_field], -1).astype(np.int64)
    mask = np.ones((n_seq, T), dtype=np.float32)
    mask[: n_seq // 2, int(T * 0.9):] = 0
    return x, y, mask


def synthetic_cp_vocabulary():
    """word2event of the same shape as the compound-word dictionary the reference trains on (class counts of
    IRL_dqn_train.py:403; event spellings as dqn_policy/testing-no-type-cp.py:56-117 parses them: 0 = ignore,
    'CONTI', 'Bar', 'Beat_<k>', 'Tempo_<bpm>', 'Note_Pitch_<p>', 'Note_Duration_<ticks>', 'Note_Velocity_<v>')."""
    tempo = [0, "CONTI"] + ["Tempo_%d" % (32 + 3 * i) for i in range(DQN_N[0] - 2)]
    roots = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]
    quals = ["M", "m", "o", "+", "7", "M7", "m7", "o7", "/o7", "sus2", "sus4"]
    chord = ([0, "CONTI"] + ["%s_%s" % (r, q) for r in roots for q in quals] + ["N_N"])[:DQN_N[1]]
    barbeat = [0, "Bar"] + ["Beat_%d" % i for i in range(DQN_N[2] - 2)]
    typ = ["EOS", "Metrical", "Note"]
    pitch = [0] + ["Note_Pitch_%d" % (22 + i) for i in range(DQN_N[4] - 1)]
    duration = [0] + ["Note_Duration_%d" % (120 * (i + 1)) for i in range(DQN_N[5] - 1)]
    velocity = [0] + ["Note_Velocity_%d" % (40 + 2 * i) for i in range(DQN_N[6] - 1)]
    cols = (tempo, chord, barbeat, typ, pitch, duration, velocity)
    assert tuple(len(c) for c in cols) == DQN_N
    return {k: dict(enumerate(c)) for k, c in zip(DQN_KEYS, cols)}


def load_dqn(path_train_data, path_dictionary, n_seq=8, T=3584, seed=1234):
    """-> (event2word, word2event), {x, y, mask}; synthetic when the files are absent."""
    if os.path.exists(path_train_data) and os.path.exists(path_dictionary):
        with open(path_dictionary, "rb") as f:
            dictionary = pickle.load(f)
        d = np.load(path_train_data)
        return dictionary, {"x": d["x"], "y": d["y"], "mask": d["mask"]}
    print("[data] %s not found: using synthetic CW tokens of the same schema" % path_train_data)
    w2e = synthetic_cp_vocabulary()
    e2w = {k: {e: i for i, e in v.items()} for k, v in w2e.items()}
    x, y, mask = _synth(n_seq, T, DQN_N, seed)
    return (e2w, w2e), {"x": x, "y": y, "mask": mask}


def ppo_vocabulary():
    """event2word of the PPO pipeline.  Unlike the CP dictionary it does not depend on the dataset: it is the fixed
    table ppo_policy/prepare_data.py:240-296 builds (value ranges per event type + <BOS>/<EOS>/<PAD>), so the
    synthetic stand-in is the real vocabulary."""
    values = {"Tempo": ["%d" % i for i in range(28, 211, 4)], "Bar": ["%d" % i for i in range(16)],
              "Position": ["%d/16" % i for i in range(16)], "Pitch": ["%d" % i for i in range(22, 108)],
              "Duration": ["%d" % i for i in range(64)], "Velocity": ["%d" % i for i in range(22)]}
    e2w = {}
    for k in PPO_KEYS:
        names = ["%s %s" % (k, v) for v in values[k]] + ["%s <BOS>" % k, "%s <EOS>" % k, "%s <PAD>" % k]
        e2w[k] = {name: i for i, name in enumerate(names)}
    assert tuple(len(e2w[k]) for k in PPO_KEYS) == PPO_N
    return e2w


def load_ppo(path_dictionary, path_train_data, n_seq=8, T=1200, seed=1234):
    if os.path.exists(path_train_data) and os.path.exists(path_dictionary):
        with open(path_dictionary, "rb") as f:
            dictionary = pickle.load(f)
        with open(path_train_data, "rb") as f:
            ds = pickle.load(f)
        return dictionary, ds
    print("[data] %s not found: using synthetic CW tokens of the same schema" % path_train_data)
    e2w = ppo_vocabulary()
    w2e = {k: {i: e for e, i in v.items()} for k, v in e2w.items()}
    x, y, mask = _synth(n_seq, T, PPO_N, seed)
    return (e2w, w2e), {"train_x": x, "train_y": y, "mask": mask}

