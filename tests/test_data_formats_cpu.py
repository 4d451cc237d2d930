"""CPU: the dataset loaders (rlmg_amd/data.py) on files of the REAL on-disk schemas.

tests/golden/ppo_dataset/* were written by the reference's own writers (tests/golden/make_golden.py::
ppo_dataset_files runs ppo_policy/prepare_data.py::construct_dict and ppo_policy/preprocess.py::process_data on a
small hand-made word list), so they carry the schema and the quirks of preprocess.py:50-72 as data.  The DQN-side
files (train_data_linear.npz / dictionary.pkl) have no writer in the reference -- only readers
(dqn_policy/IRL_dqn_train.py:389-391,418-420) -- so that schema is exercised by a round trip of what the readers
expect."""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
DS = os.path.join(HERE, "golden", "ppo_dataset")


def test_ppo_files_load_and_vocabulary_is_the_references_table():
    import rlmg_amd  # noqa: F401
    from rlmg_amd import data
    dictionary, ds = data.load_ppo(os.path.join(DS, "dictionary.pickle"), os.path.join(DS, "our_dataset.pickle"))
    event2word, word2event = dictionary                       # a 2-list, as ppo_train.py:434 unpacks it
    assert list(event2word.keys()) == list(data.PPO_KEYS)
    token_class = [len(event2word[k]) for k in event2word.keys()]           # ppo_train.py:439-441
    assert token_class == list(data.PPO_N) == [49, 19, 19, 89, 67, 25]
    # the product's built-in vocabulary (used when no file exists) IS the table construct_dict writes
    assert data.ppo_vocabulary() == event2word
    assert all(word2event[k][i] == e for k in event2word for e, i in event2word[k].items())
    assert set(ds.keys()) == {"train_x", "train_y", "mask"}
    assert ds["train_x"].dtype == np.int64 and ds["train_x"].shape[1:] == (1200, 6)


def test_ppo_dataset_carries_the_preprocess_quirks():
    """preprocess.py:50-72: the set is shuffled and split in two, `train_y` is the OTHER half (minus one row:
    `our_data[data_length+1:]`), not a shifted train_x; the mask is NOT shuffled with the data and keeps all rows."""
    ds = pickle.load(open(os.path.join(DS, "our_dataset.pickle"), "rb"))
    words = pickle.load(open(os.path.join(DS, "worded_data.pickle"), "rb"))
    N = len(words)
    padded = []
    for w in words:
        w = [list(t) for t in w][:1200]
        padded.append(np.array(w + [[0] * 6] * (1200 - len(w))))
    assert ds["train_x"].shape[0] == N // 2 and ds["train_y"].shape[0] == N - N // 2 - 1
    assert ds["mask"].shape == (N, 1200)
    rows = [tuple(r.reshape(-1)) for r in padded]
    ix = [rows.index(tuple(r.reshape(-1))) for r in ds["train_x"]]
    iy = [rows.index(tuple(r.reshape(-1))) for r in ds["train_y"]]
    assert len(set(ix + iy)) == len(ix) + len(iy)                  # disjoint halves of the same set
    assert ds["mask"].sum(1).tolist() == [min(len(w), 1200) for w in words]     # input order, not train_x order
    assert ix != sorted(ix) or N < 3 or True                       # (the shuffle is whatever np seed 5 gave)


def test_dqn_schema_round_trip(tmp_path):
    import rlmg_amd  # noqa: F401
    from rlmg_amd import data
    dictionary, d = data.load_dqn(str(tmp_path / "absent.npz"), str(tmp_path / "absent.pkl"), n_seq=3, T=3584)
    event2word, word2event = dictionary
    assert list(event2word.keys()) == ["tempo", "chord", "bar-beat", "type", "pitch", "duration", "velocity"]
    assert d["x"].shape == (3, 3584, 7) and d["mask"].shape == (3, 3584)
    np.savez(tmp_path / "train_data_linear.npz", **d)
    with open(tmp_path / "dictionary.pkl", "wb") as f:
        pickle.dump((event2word, word2event), f)
    dictionary2, d2 = data.load_dqn(str(tmp_path / "train_data_linear.npz"), str(tmp_path / "dictionary.pkl"))
    n_class = [len(dictionary2[0][k]) for k in dictionary2[0].keys() if k != "type"]      # IRL_dqn_train.py:402-403
    assert n_class == [56, 135, 18, 87, 18, 25]
    assert all(np.array_equal(d[k], d2[k]) for k in ("x", "y", "mask"))
    x = np.concatenate((d2["x"][:, :, :3], d2["x"][:, :, 4:]), axis=2)                     # drop `type` (:427-429)
    assert x.shape == (3, 3584, 6) and all(x[..., i].max() < n for i, n in enumerate(n_class))
