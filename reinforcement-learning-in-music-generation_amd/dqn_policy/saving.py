"""Drop-in for /root/reference/dqn_policy/saving.py::Saver: the experiment log
`<exp_dir>/log.txt`, one line per summary in the reference's format

    '{key:10s} | {val:.10f} | {step:10d} | {seconds since construction}'      (float values)
    '{key:10s} | {val} | {step:10d} | {seconds}'                              (anything else)

so that existing log parsers (saving.py:108-121 splits on ' | ') keep working, plus save_model / load_model
with the reference's file names (<name>.pt, <name>_params.pt, <name>_opt.pt).  Plotting (`make_loss_report`,
matplotlib) is not rebuilt.  Uses its own logger + file handler instead of logging.basicConfig, so that importing
it does not reconfigure the root logger of the host program.
"""
import logging
import os
import time

import torch


class Saver(object):
    def __init__(self, exp_dir, mode="w"):
        self.exp_dir = exp_dir
        self.init_time = time.time()
        self.global_step = 0
        os.makedirs(exp_dir, exist_ok=True)
        self.logger = logging.getLogger("training monitor:" + os.path.abspath(exp_dir))
        self.logger.setLevel(logging.DEBUG)
        self.logger.propagate = False
        for h in list(self.logger.handlers):
            self.logger.removeHandler(h)
            h.close()
        handler = logging.FileHandler(os.path.join(exp_dir, "log.txt"), mode=mode)
        handler.setFormatter(logging.Formatter("%(message)s"))
        self.logger.addHandler(handler)

    def add_summary_msg(self, msg):
        self.logger.debug(msg)

    def add_summary(self, key, val, step=None, cur_time=None):
        if cur_time is None:
            cur_time = time.time() - self.init_time
        if step is None:
            step = self.global_step
        if isinstance(val, float):
            msg = "{:10s} | {:.10f} | {:10d} | {}".format(key, val, step, cur_time)
        else:
            msg = "{:10s} | {} | {:10d} | {}".format(key, val, step, cur_time)
        self.logger.debug(msg)

    def save_model(self, model, optimizer=None, outdir=None, name="model"):
        if outdir is None:
            outdir = self.exp_dir
        print(" [*] saving model to {}, name: {}".format(outdir, name))
        torch.save(model, os.path.join(outdir, name + ".pt"))
        torch.save(model.state_dict(), os.path.join(outdir, name + "_params.pt"))
        if optimizer is not None:
            torch.save(optimizer.state_dict(), os.path.join(outdir, name + "_opt.pt"))

    def load_model(self, path_exp, device="cpu", name="model.pt"):
        path_pt = os.path.join(path_exp, name)
        print(" [*] restoring model from", path_pt)
        return torch.load(path_pt, map_location=torch.device(device), weights_only=False)

    def global_step_increment(self):
        self.global_step += 1

    def close(self):
        for h in list(self.logger.handlers):
            h.flush()
            self.logger.removeHandler(h)
            h.close()
