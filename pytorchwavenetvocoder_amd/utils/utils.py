# -*- coding: utf-8 -*-
"""Host-side helpers with the API of the reference's ``wavenet_vocoder/utils/utils.py``:
``check_hdf5 / read_hdf5 / shape_hdf5 / write_hdf5`` (utils.py:18-139), ``find_files`` / ``read_txt``
(:142-162), ``BackgroundGenerator`` / ``background`` (:165-217), ``extend_time`` (:220-242).

Differences that matter:
  * ``h5py`` is optional.  When it is not installed (or the file is not HDF5) the same four
    functions read/write a ``.npz`` container keyed by the dataset path, so the training CLI and
    its tests work in a minimal environment; with h5py present the files are ordinary HDF5 and
    interchangeable with the reference's.
  * ``background(max_prefetch=N)`` really prefetches N items (the reference drops the argument,
    utils.py:216, so its queue depth is always 1).
"""
import fnmatch
import logging
import os
import sys
import threading
import zipfile
from queue import Full, Queue

import numpy as np

try:  # optional
    import h5py
except ImportError:  # pragma: no cover
    h5py = None

__all__ = ["check_hdf5", "read_hdf5", "shape_hdf5", "write_hdf5", "find_files", "read_txt",
           "BackgroundGenerator", "background", "extend_time", "make_feat_transform"]


def _is_npz(path):
    return zipfile.is_zipfile(path)


def _npz_load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def _key(hdf5_path):
    return hdf5_path.strip("/").replace("/", "__")


def check_hdf5(hdf5_name, hdf5_path):
    """True when dataset ``hdf5_path`` exists in file ``hdf5_name``."""
    if not os.path.exists(hdf5_name):
        return False
    if _is_npz(hdf5_name):
        return _key(hdf5_path) in _npz_load(hdf5_name)
    if h5py is None:
        return False
    with h5py.File(hdf5_name, "r") as f:
        return hdf5_path in f


def read_hdf5(hdf5_name, hdf5_path):
    """Read dataset ``hdf5_path`` of file ``hdf5_name`` (exits like the reference on a miss)."""
    if not os.path.exists(hdf5_name):
        logging.error("there is no such a hdf5 file (%s)." % hdf5_name)
        sys.exit(1)
    if _is_npz(hdf5_name):
        data = _npz_load(hdf5_name)
        if _key(hdf5_path) not in data:
            logging.error("there is no such a data in hdf5 file. (%s)" % hdf5_path)
            sys.exit(1)
        return data[_key(hdf5_path)]
    if h5py is None:
        logging.error("h5py is not installed and %s is not an .npz container." % hdf5_name)
        sys.exit(1)
    with h5py.File(hdf5_name, "r") as f:
        if hdf5_path not in f:
            logging.error("there is no such a data in hdf5 file. (%s)" % hdf5_path)
            sys.exit(1)
        return f[hdf5_path][()]


def shape_hdf5(hdf5_name, hdf5_path):
    """Shape of a dataset."""
    if not check_hdf5(hdf5_name, hdf5_path):
        logging.error("there is no such a file or dataset")
        sys.exit(1)
    return tuple(np.shape(read_hdf5(hdf5_name, hdf5_path)))


def write_hdf5(hdf5_name, hdf5_path, write_data, is_overwrite=True):
    """Write ``write_data`` as dataset ``hdf5_path`` of file ``hdf5_name``."""
    write_data = np.array(write_data)
    folder_name, _ = os.path.split(hdf5_name)
    if len(folder_name) != 0 and not os.path.exists(folder_name):
        os.makedirs(folder_name)
    use_npz = h5py is None or (os.path.exists(hdf5_name) and _is_npz(hdf5_name))
    if use_npz:
        data = _npz_load(hdf5_name) if os.path.exists(hdf5_name) else {}
        if _key(hdf5_path) in data and not is_overwrite:
            logging.error("dataset in hdf5 file already exists.")
            sys.exit(1)
        data[_key(hdf5_path)] = write_data
        with open(hdf5_name, "wb") as f:  # keep the given file name (np.savez would append .npz)
            np.savez(f, **data)
        return
    with h5py.File(hdf5_name, "a") as f:
        if hdf5_path in f:
            if not is_overwrite:
                logging.error("dataset in hdf5 file already exists.")
                sys.exit(1)
            logging.warning("dataset in hdf5 file already exists. recreate dataset in hdf5.")
            del f[hdf5_path]
        f.create_dataset(hdf5_path, data=write_data)


def find_files(directory, pattern="*.wav", use_dir_name=True):
    """Recursively list files matching ``pattern`` (relative names when use_dir_name=False)."""
    files = []
    for root, _dirs, names in os.walk(directory, followlinks=True):
        for name in fnmatch.filter(names, pattern):
            files.append(os.path.join(root, name))
    if not use_dir_name:
        files = [f.replace(directory + "/", "") for f in files]
    return files


def read_txt(file_list):
    """One entry per line."""
    with open(file_list, "r") as f:
        return [line.rstrip("\n") for line in f.readlines()]


class _ProducerError(object):
    """Wrapper that carries an exception of the producer thread to the consumer."""

    def __init__(self, error):
        self.error = error


class BackgroundGenerator(threading.Thread):
    """Runs ``generator`` in a daemon thread and hands items over through a bounded queue."""

    def __init__(self, generator, max_prefetch=1):
        threading.Thread.__init__(self)
        self.queue = Queue(max_prefetch)
        self.generator = generator
        self._closed = False
        self.daemon = True
        self.start()

    def run(self):
        # a producer that dies must not leave the consumer blocked in queue.get(): its exception travels through the
        # queue and is re-raised by next(); a normal end hands over the None sentinel
        try:
            for item in self.generator:
                if not self._put(item):
                    return
        except BaseException as e:  # noqa: B902  (re-raised in the consumer thread)
            self._put(_ProducerError(e))
            return
        self._put(None)

    def _put(self, item):
        while not self._closed:
            try:
                self.queue.put(item, timeout=0.2)
                return True
            except Full:
                continue
        return False

    def close(self, timeout=5.0):
        """Stop the producer thread (it exits at its next hand-over), wait for it, and close the generator it was running
        (its ``finally`` clauses -- e.g. the window slicer's worker pool -- run now, not at interpreter exit)."""
        self._closed = True
        if threading.current_thread() is not self:
            try:                      # free a slot: a producer blocked in put() sees _closed at its next time-out anyway
                self.queue.get_nowait()
            except Exception:  # noqa: BLE001
                pass
            self.join(timeout)
            if not self.is_alive():
                try:
                    self.generator.close()
                except Exception:  # noqa: BLE001 -- (a generator that is still executing cannot be closed: left to the daemon flag)
                    pass

    def next(self):
        item = self.queue.get()
        if item is None:
            raise StopIteration
        if isinstance(item, _ProducerError):
            self._closed = True
            raise item.error
        return item

    __next__ = next

    def __iter__(self):
        return self


class background(object):
    """Decorator: ``@background(max_prefetch=16)`` turns a generator function into a prefetching one."""

    def __init__(self, max_prefetch=1):
        self.max_prefetch = max_prefetch

    def __call__(self, gen):
        def bg_generator(*args, **kwargs):
            return BackgroundGenerator(gen(*args, **kwargs), max_prefetch=self.max_prefetch)
        return bg_generator


def extend_time(feats, upsampling_factor):
    """(T, D) -> (upsampling_factor * T, D) by repeating every frame."""
    return np.repeat(np.asarray(feats), upsampling_factor, axis=0).astype(np.float64)


def make_feat_transform(mean, scale):
    """``StandardScaler.transform`` with given statistics (reference train.py:463-465,468-469): a float copy of the
    features, ``-= mean`` and ``/= scale`` IN PLACE -- i.e. in the features' own dtype, each step rounded to it, which is
    what the reference's float32 features get from float64 statistics (bit-equal: tests/test_train_cli.py slicer golden)."""
    mean, scale = np.asarray(mean), np.asarray(scale)

    def transform(x):
        x = np.array(x, dtype=x.dtype if np.issubdtype(np.asarray(x).dtype, np.floating) else np.float64, copy=True)
        x -= mean
        x /= scale
        return x
    return transform
