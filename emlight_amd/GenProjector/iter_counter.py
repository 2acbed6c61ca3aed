"""Iteration bookkeeping and resume of the projector trainer (reference ``GenProjector/iter_counter.py``).

Counts SAMPLES like the reference (``total_steps_so_far`` advances by the global batch per iteration), decides when to
print / save, and persists ``(epoch, samples into the epoch)`` as ``<checkpoints_dir>/<name>/iter.txt`` so that
``--continue_train`` resumes where ``latest_net_{G,D}.pth`` was written (``iter_counter.py:19-29,57-64``)."""
import os
import time

import numpy as np


class IterationCounter:
    def __init__(self, checkpoints_dir, name, dataset_size, batch_size, niter, niter_decay=0, continue_train=False,
                 print_freq=100, save_latest_freq=5000, save_epoch_freq=10):
        self.dataset_size, self.batch_size = dataset_size, batch_size
        self.print_freq, self.save_latest_freq, self.save_epoch_freq = print_freq, save_latest_freq, save_epoch_freq
        self.first_epoch, self.epoch_iter = 1, 0
        self.total_epochs = niter + niter_decay
        self.iter_record_path = os.path.join(checkpoints_dir, name, "iter.txt")
        if continue_train:
            try:
                self.first_epoch, self.epoch_iter = (int(v) for v in np.loadtxt(self.iter_record_path, delimiter=",", dtype=int))
                print("Resuming from epoch %d at iteration %d" % (self.first_epoch, self.epoch_iter))
            except (OSError, ValueError):
                print("Could not load iteration record at %s. Starting from beginning." % self.iter_record_path)
        self.total_steps_so_far = (self.first_epoch - 1) * dataset_size + self.epoch_iter
        self.current_epoch = self.first_epoch
        self.time_per_iter = 0.0

    def training_epochs(self):
        return range(self.first_epoch, self.total_epochs + 1)

    def record_epoch_start(self, epoch):
        """The first resumed epoch continues from the recorded offset; later epochs start at 0."""
        self.epoch_start_time = self.last_iter_time = time.time()
        if epoch != self.first_epoch:
            self.epoch_iter = 0
        self.current_epoch = epoch

    def record_one_iteration(self):
        now = time.time()
        self.time_per_iter = (now - self.last_iter_time) / self.batch_size
        self.last_iter_time = now
        self.total_steps_so_far += self.batch_size
        self.epoch_iter += self.batch_size

    def record_epoch_end(self, write=True):
        self.time_per_epoch = time.time() - self.epoch_start_time
        if write and self.current_epoch % self.save_epoch_freq == 0:
            np.savetxt(self.iter_record_path, (self.current_epoch + 1, 0), delimiter=",", fmt="%d")

    def record_current_iter(self):
        np.savetxt(self.iter_record_path, (self.current_epoch, self.epoch_iter), delimiter=",", fmt="%d")

    def needs_saving(self):
        return (self.total_steps_so_far % self.save_latest_freq) < self.batch_size

    def needs_printing(self):
        return (self.total_steps_so_far % self.print_freq) < self.batch_size
