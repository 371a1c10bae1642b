"""Decode worker of `BatchGenerator.generate(workers=N)`: a plain subprocess (`python -m fcn8s_tensorflow_amd._feed_worker`)
that reads length-prefixed pickled sample tasks from stdin, decodes + augments them (batch_generator._process_sample) and
writes the pixels straight into a /dev/shm batch buffer the parent mapped; only a few bytes of acknowledgement travel back
through the pipe.  Deliberately not `multiprocessing`: spawn / forkserver re-import the user's training script (the reference's
scripts and notebook cells have no `if __name__ == "__main__"` guard) and fork would duplicate a process that holds a HIP context.
"""
import pickle
import struct
import sys

import numpy as np


def _read(stream):
    head = stream.read(4)
    if len(head) < 4:
        return None
    (n,) = struct.unpack("<I", head)
    return pickle.loads(stream.read(n))


def _write(stream, obj):
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    stream.write(struct.pack("<I", len(blob)))
    stream.write(blob)
    stream.flush()


def main():
    from fcn8s_tensorflow_amd.batch_generator import _process_sample
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    maps = {}
    while True:
        msg = _read(inp)
        if msg is None:
            return
        task, dest = msg
        try:
            image, gt = _process_sample(task)
            if dest is not None:
                path, img_off, img_shape, gt_off, gt_shape = dest
                ok = tuple(image.shape) == tuple(img_shape) and image.dtype == np.uint8 and \
                    (gt is None or (gt_shape is not None and tuple(gt.shape) == tuple(gt_shape) and gt.dtype == np.uint8))
                if ok:
                    if path not in maps:
                        # the parent alternates between two slot files per batch geometry and makes new ones when the geometry
                        # changes: keep the two most recent mappings
                        while len(maps) >= 2:
                            maps.pop(next(iter(maps)))
                        maps[path] = np.memmap(path, dtype=np.uint8, mode="r+")
                    buf = maps[path]
                    buf[img_off:img_off + image.size] = image.reshape(-1)
                    if gt is not None:
                        buf[gt_off:gt_off + gt.size] = gt.reshape(-1)
                    _write(out, ("shm",))
                    continue
            _write(out, ("arr", image, gt))
        except BaseException as e:                        # reported to the parent, which re-raises
            _write(out, ("err", repr(e)))


if __name__ == "__main__":
    main()
