"""One rank of tests/test_fake_rccl_gpu.py: a process of its own on GPU 0 whose libfcn8s_hip.so talks to its peers through the
shared-memory stand-in for librccl (FCN8S_RCCL_LIBRARY, set by the test).  No torch.distributed anywhere: the 128-byte unique id
travels through a file, as include/fcn8s_hip.h says any means will do.

    python tests/fake_rccl/worker.py <mode> <rank> <world> <id file> <out file>
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SMALL = (8, 16, 32, 64, 64, 128, 128)


def unique_id(L, rank, path):
    if rank == 0:
        buf = C.create_string_buffer(L.COMM_ID_BYTES)
        L.check(L.lib.fcn8s_comm_unique_id(buf, L.COMM_ID_BYTES))
        with open(path + ".tmp", "wb") as f:
            f.write(buf.raw)
        os.replace(path + ".tmp", path)
        return buf.raw
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 120, "rank 0 never wrote the unique id"
        time.sleep(0.01)
    return open(path, "rb").read()


def main():
    mode, rank, world, idfile, outfile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import _lib as L
    from oracle import fcn8s_oracle as orc      # checker only: the synthetic parameters of the other GPU tests
    from tests.test_facade_gpu import gen
    res = {"rank": rank, "mode": mode}
    e = Engine(20, widths=SMALL, device_id=0, seed=7)
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    if rank != 0:                               # rank 0's parameters must arrive through the broadcast, not through the shared seed
        P = {k: np.zeros_like(v) for k, v in P.items()}
    e.set_params(P)
    img, lab = next(gen(2 * world, 32, 64, 4, onehot=False))
    sl = slice(2 * rank, 2 * rank + 2)

    if mode == "version":
        try:
            e.comm_init_native(unique_id(L, 0, idfile), 0, 1)
            res["raised"] = False
        except L.Fcn8sError as ex:
            res["raised"], res["msg"] = True, str(ex)
        json.dump(res, open(outfile, "w")); e.close(); return

    e.comm_init_native(unique_id(L, rank, idfile), rank, world)
    info = e.comm_info()
    assert info["world"] == world and info["rank"] == rank and info["rccl_version"] // 10000 == 2, info
    e.broadcast_params(0)

    if mode == "step":
        loss, step = e.train_step(img[sl], lab[sl], 1e-2, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
        loss2, step2 = e.train_step(img[sl], lab[sl], 1e-2, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)     # buckets re-used: pending flags, events
        params2 = e.flat_params.cpu().numpy().copy()
        # ... and the C entry point a plain caller uses: fcn8s_train_step trains data-parallel on its own when the communicator has > 1 rank (TF-Adam)
        ka, pi, dt, pl, where, nhw = e._inputs(img[sl], lab[sl])
        st = C.c_int64(0); lo = C.c_float(0)
        L.check(L.lib.fcn8s_train_step(e.h, pi, dt, pl, 2, 32, 64, 1e-3, 1.0, 0.0, where, C.byref(lo), C.byref(st)), e.h)
        e.metrics_reset(); e.eval_step(img[sl], lab[sl]); e.metrics_allreduce()
        cm, ls, lc = e.metrics_raw()
        np.savez(outfile + ".npz", params=e.flat_params.cpu().numpy(), params2=params2)
        res.update(loss=loss, step=step, loss2=loss2, step2=step2, step3=int(st.value), conf_sum=int(cm.sum()), loss_count=int(lc))
        e.comm_destroy()
        e.close()
    elif mode in ("stall_update", "async_error"):
        # a peer that never arrives (stall) / a failure RCCL reports asynchronously: the step must END with FCN8S_ERR_RCCL, raised by the
        # call that would have applied the update, and the parameters must not have been touched by it
        timeout_ms = int(os.environ.get("TEST_COMM_TIMEOUT_MS", "1500"))
        e.set_option("comm_timeout_ms", timeout_ms)
        before = e.flat_params.cpu().numpy().copy()
        t0 = time.time()
        try:
            e.train_step(img[sl], lab[sl], 1e-2, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
            res["raised"] = False
        except L.Fcn8sError as ex:
            res["raised"], res["msg"] = True, str(ex)
        res["elapsed_s"] = time.time() - t0
        res["step_after"] = e.global_step
        res["params_untouched"] = bool(np.array_equal(before, e.flat_params.cpu().numpy()))
        try:                                    # every later call says so too
            e.train_step(img[sl], lab[sl], 1e-2, keep_prob=1.0, optimizer=L.OPT_SGD_MOMENTUM)
            res["second_raised"] = False
        except L.Fcn8sError as ex:
            res["second_raised"], res["second_msg"] = True, str(ex)
        try:
            e.metrics_allreduce(); res["metrics_raised"] = False
        except L.Fcn8sError:
            res["metrics_raised"] = True
        try:
            e.comm_destroy(); res["destroy_raised"] = False
        except L.Fcn8sError:
            res["destroy_raised"] = True
        # the model is usable again on its own
        loss, _ = e.train_step(img[sl], lab[sl], 1e-2, keep_prob=1.0, optimizer=L.OPT_SGD_MOMENTUM)
        res["alone_loss_finite"] = bool(np.isfinite(loss))
        e.close()
    elif mode == "stall_destroy":
        # the all-reduces are in flight against a peer that never arrives and the caller goes straight to fcn8s_destroy: it must return
        # (FCN8S_ERR_RCCL, reason in fcn8s_last_error(NULL)) after about comm_timeout_ms, not hang in a device synchronisation
        e.set_option("comm_timeout_ms", int(os.environ.get("TEST_COMM_TIMEOUT_MS", "1500")))
        ka, pi, dt, pl, where, nhw = e._inputs(img[sl], lab[sl])
        L.check(L.lib.fcn8s_forward_loss(e.h, pi, dt, pl, 2, 32, 64, 1.0, 0.0, where), e.h)
        nb = e.num_buckets
        for b in range(nb):
            L.check(L.lib.fcn8s_backward_bucket(e.h, b), e.h)
            for r in range(nb):
                if int(L.lib.fcn8s_bucket_complete_after(e.h, r)) == b:
                    L.check(L.lib.fcn8s_allreduce_bucket(e.h, r), e.h)
        t0 = time.time()
        rc = L.lib.fcn8s_destroy(e.h); e.h = None
        res["elapsed_s"] = time.time() - t0
        res["rc"] = int(rc)
        msg = L.lib.fcn8s_last_error(None)
        res["msg"] = msg.decode() if msg else ""
        res["err_rccl"] = int(L.ERR_RCCL)
    elif mode == "metrics_stall":
        # rank 1 never calls the metrics all-reduce: rank 0's call must end with FCN8S_ERR_RCCL (the watchdog covers this collective too)
        e.set_option("comm_timeout_ms", int(os.environ.get("TEST_COMM_TIMEOUT_MS", "1500")))
        e.metrics_reset(); e.eval_step(img[sl], lab[sl])
        t0 = time.time()
        if rank == 0:
            try:
                e.metrics_allreduce(); res["raised"] = False
            except L.Fcn8sError as ex:
                res["raised"], res["msg"] = True, str(ex)
        else:
            time.sleep(float(os.environ.get("TEST_COMM_TIMEOUT_MS", "1500")) / 1000.0 + 3.0)
        res["elapsed_s"] = time.time() - t0
        try:
            e.comm_destroy()
        except L.Fcn8sError:
            pass
        e.close()
    json.dump(res, open(outfile, "w"))


if __name__ == "__main__":
    main()
