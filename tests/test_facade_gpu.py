"""The FCN8s facade (drop-in for fcn8s_tensorflow.FCN8s) end to end on the GPU: training loop,
evaluation, prediction, save / resume, the reference's validation errors, and the RCCL code path
with a one-rank process group."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fcn8s_oracle as orc  # noqa: E402  (checker only)

SMALL = (8, 16, 32, 64, 64, 128, 128)


def gen(n, h, w, seed, onehot=True):
    rng = np.random.default_rng(seed)
    while True:
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        lab = rng.integers(0, 20, (n, h, w), dtype=np.uint8)
        lab[:, : h // 2] = (img[:, : h // 2, :, 0] > 127).astype(np.uint8) * 5      # learnable structure
        yield img, (orc.one_hot(lab, 20) if onehot else lab)


def make(**kw):
    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    return FCN8s(vgg16_dir='synthetic:3', num_classes=20, widths=SMALL, **kw)


def test_constructor_validation():
    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    with pytest.raises(ValueError, match="You must provide either both"):
        FCN8s()
    with pytest.raises(ValueError):
        FCN8s(vgg16_dir='/nonexistent/dir', num_classes=20, widths=SMALL)


def test_train_evaluate_predict_save_resume(tmp_path, capsys):
    m = make()
    g = gen(2, 32, 64, 0)
    with pytest.raises(ValueError, match="`eval_dataset` must be one of"):
        m.train(g, 1, 1, lambda s: 1e-3, eval_dataset='test')
    with pytest.raises(ValueError, match="not a valid metric"):
        m.train(g, 1, 1, lambda s: 1e-3, metrics={'f1'})
    with pytest.raises(ValueError, match="You are trying to monitor"):
        m.train(g, 1, 1, lambda s: 1e-3, monitor='mean_iou')
    with pytest.raises(ValueError, match="a `val_generator` and `val_steps` must be passed"):
        m.train(g, 1, 1, lambda s: 1e-3, eval_dataset='val')

    m.save(str(tmp_path), 'saved_model')                      # nothing trained yet
    assert "Abort: Nothing to save" in capsys.readouterr().out

    lrs = []
    def schedule(step):
        lrs.append(step)
        return 1e-3 if step < 4 else 5e-4
    m.train(g, epochs=2, steps_per_epoch=3, learning_rate_schedule=schedule, keep_prob=0.5, l2_regularization=1e-4,
            eval_dataset='val', eval_frequency=1, val_generator=gen(2, 32, 64, 1), val_steps=2,
            metrics={'loss', 'mean_iou', 'accuracy'}, save_during_training=True, save_dir=str(tmp_path),
            save_best_only=True, save_frequency=1, saver='saved_model', monitor='loss',
            summaries_dir=str(tmp_path / 'tb'), summaries_name='run', training_loss_display_averaging=3)
    assert m.g_step == 6 and lrs == [0, 1, 2, 3, 4, 5, 6]
    assert m.metric_names == ['loss', 'mean_iou', 'accuracy'] and len(m.metric_values) == 3
    assert np.isfinite(m.training_loss) and m.variables_updated is False
    saved = [d for d in os.listdir(tmp_path) if d.startswith('saved_model__(globalstep-')]   # save_name='' -> 'saved_model_' + '' + '_(globalstep-..', as in the reference
    assert saved and all('(eval_on_val_dataset)' in d and '(trainloss-' in d and '(mean_iou-' in d for d in saved)
    assert os.path.isfile(tmp_path / 'tb' / 'run' / 'scalars.jsonl')
    # TensorBoard event files like the reference's two FileWriters write (fcn8s_tensorflow.py:331-366, :531-535)
    from glob import glob
    from fcn8s_tensorflow_amd import tf_events
    (tr,) = glob(str(tmp_path / 'tb' / 'run' / 'events.out.tfevents.*'))
    evs = tf_events.read_events(tr)
    assert evs[0]['file_version'] == 'brain.Event:2' and len(evs) >= 2
    rec = evs[1]
    assert {'total_loss', 'learning_rate', 'fc6/kernel/mean', 'fc6/kernel/stddev_1', 'conv3_3/bias/max', 'pool3_1x1/kernel/min'} <= set(rec['scalars'])
    assert len(rec['histograms']) == 20 and rec['histograms']['fc7/kernel/histogram']['num'] == float(np.prod(m.engine.specs['fc7/weights'][0]))
    (ev,) = glob(str(tmp_path / 'tb' / 'run_eval' / 'events.out.tfevents.*'))
    assert set(tf_events.read_events(ev)[1]['scalars']) == {'mean_loss', 'mean_iou', 'accuracy'}

    m.evaluate(gen(2, 32, 64, 2), num_batches=2, metrics={'loss', 'mean_iou'}, dataset='val')
    assert m.metric_names == ['loss', 'mean_iou'] and m.eval_dataset == 'val'
    out = capsys.readouterr().out
    assert 'loss: ' in out and 'mean_iou: ' in out
    with pytest.raises(ValueError, match="`dataset` must be either"):
        m.evaluate(g, 1, dataset='test')

    img = next(gen(1, 32, 64, 9))[0][0]
    pred = m.predict([img])                                    # a Python list of HWC arrays, as the reference allows
    assert pred.shape == (1, 32, 64) and pred.dtype == np.int64
    sm = m.predict([img], argmax=False)
    assert sm.shape == (1, 32, 64, 20) and np.allclose(sm.sum(-1), 1, atol=1e-5)
    np.testing.assert_array_equal(np.argmax(sm, -1), pred)

    # train_saver + resume: parameters, Adam slots and global_step round-trip
    m.variables_updated = True
    m.save(str(tmp_path / 'ck'), 'train_saver', name='x', include_metrics=False)
    name = m.last_saved_model_name
    assert name.startswith('saved_model_x_(globalstep-6)_(trainloss-')
    with pytest.raises(ValueError, match="Unexpected value for `saver`"):
        m.variables_updated = True
        m.save(str(tmp_path), 'pickle')
    params = m.engine.get_params(); mom = m.engine.get_opt_state()
    m.close()
    assert "The session has been closed." in capsys.readouterr().out

    m2 = make(variables_load_dir=str(tmp_path / 'ck' / name / 'variables'))
    assert m2.engine.global_step == 6
    for k, v in m2.engine.get_params().items():
        np.testing.assert_array_equal(v, params[k])
    np.testing.assert_array_equal(m2.engine.get_opt_state()[0], mom[0])
    np.testing.assert_array_equal(m2.predict([img]), pred)
    m2.close()

    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    m3 = FCN8s(model_load_dir=str(tmp_path / saved[-1]), tags=['default'])
    assert m3.num_classes == 20 and m3.engine.global_step in (3, 6)
    m3.close()


def test_predict_and_save(tmp_path):
    from PIL import Image
    m = make()
    src = tmp_path / 'in'; src.mkdir()
    rng = np.random.default_rng(0)
    for i in range(2):
        Image.fromarray(rng.integers(0, 256, (32, 64, 3), dtype=np.uint8)).save(src / ('f%d.png' % i))
    m.predict_and_save(str(tmp_path / 'out'), str(src), {c: (0, 255, 0, 127) for c in range(20)}, include_unprocessed_image=True)
    outs = sorted(os.listdir(tmp_path / 'out'))
    assert outs == ['f0.png', 'f1.png'] and Image.open(tmp_path / 'out' / 'f0.png').size == (64, 64)
    m.close()


def test_rccl_path_with_one_rank_process_group():
    """Exercises the split-phase train step (forward_loss / backward_bucket / all-reduce / update)
    under an initialised nccl (= RCCL) process group; 8-GPU runs are the driver's."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from fcn8s_tensorflow_amd.engine import Engine
        from fcn8s_tensorflow_amd import _lib as L
        P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
        img, lab = next(gen(2, 32, 64, 4, onehot=False))
        a = Engine(20, widths=SMALL); a.set_params(P)
        b = Engine(20, widths=SMALL); b.set_params(P)
        la, _ = a.train_step(img, lab, 1e-3, keep_prob=1.0)                               # fused C call
        lb, _ = b.train_step(img, lab, 1e-3, keep_prob=1.0, optimizer=L.OPT_SGD_MOMENTUM)   # split-phase path
        assert abs(la - lb) < 1e-6
        ga, gb = a.get_grads(), b.get_grads()
        for k in ga:
            assert np.abs(ga[k] - gb[k]).max() <= 1e-4 * (np.abs(ga[k]).max() + 1e-30), k   # only atomics order differs
        a.metrics_reset(); a.eval_step(img, lab); a.metrics_allreduce()
        assert a.metrics_raw()[0].sum() == lab.size
        a.close(); b.close()
    finally:
        dist.destroy_process_group()


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # two ranks share the one GPU of the test box
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import _lib as L
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = next(gen(4, 32, 64, 4, onehot=False))
    e = Engine(20, widths=SMALL, device_id=0, seed=7)
    e.set_params(P)
    e.broadcast_params(0)
    sl = slice(2 * rank, 2 * rank + 2)
    loss, step = e.train_step(img[sl], lab[sl], 1e-2, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
    e.metrics_reset(); e.eval_step(img[sl], lab[sl]); e.metrics_allreduce()
    out[rank] = (e.flat_params.cpu().numpy(), loss, step, int(e.metrics_raw()[0].sum()), e.metrics_raw()[2])
    e.close()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_big_batch_step():
    """world_size 2 through the real Engine DP path (split-phase backward, bucketed async all-reduce, 1/world folded
    into the optimizer kernel).  The test box has one GPU, so both ranks use it and the collective runs over gloo;
    on the 8-GPU node the same code runs over RCCL (backend 'nccl')."""
    import socket
    import torch.multiprocessing as mp
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import _lib as L
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    p0, p1 = out[0][0], out[1][0]
    np.testing.assert_array_equal(p0, p1)                           # replicas stay bit-identical
    assert out[0][2] == out[1][2] == 1
    assert out[0][3] == 4 * 32 * 64 and out[0][4] == 2              # confusion matrix / loss samples summed over ranks
    # reference: one process, the whole batch of 4
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = next(gen(4, 32, 64, 4, onehot=False))
    e = Engine(20, widths=SMALL); e.set_params(P)
    before = e.flat_params.cpu().numpy().copy()
    e.train_step(img, lab, 1e-2, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
    ref = e.flat_params.cpu().numpy()
    e.close()
    upd_ref, upd_dp = ref - before, p0 - before
    assert np.abs(upd_ref).max() > 0
    assert np.abs(upd_dp - upd_ref).max() <= 2e-3 * np.abs(upd_ref).max()


def _dp_traj_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import _lib as L
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=5.0, bias_std=0.05)
    e = Engine(20, widths=SMALL, device_id=0, seed=7)
    e.set_params(P)
    e.broadcast_params(0)
    e.replica_check_every = 2
    g = gen(4, 32, 64, 4, onehot=False)
    losses = []
    for t in range(6):
        img, lab = next(g)
        sl = slice(2 * rank, 2 * rank + 2)
        loss, step = e.train_step(img[sl], lab[sl], 0.1, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
        losses.append(loss)
    out[rank] = (e.flat_params.cpu().numpy(), losses, step)
    e.close()
    dist.destroy_process_group()


def test_two_rank_free_running_training_equals_the_big_batch_run():
    """Six SGD+momentum steps, free-running: two data-parallel ranks (half a batch each, gradients all-reduced in four buckets, 1/world in
    the update, the replica guard firing every second step) against one process stepping on the whole batches.  The replicas stay
    bit-identical to each other, the mean of their losses is the big batch's loss, and after six steps the parameters have moved the same way."""
    import socket
    import torch.multiprocessing as mp
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import _lib as L
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_dp_traj_worker, args=(2, port, out), nprocs=2, join=True)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert out[0][2] == out[1][2] == 6
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=5.0, bias_std=0.05)
    e = Engine(20, widths=SMALL); e.set_params(P)
    before = e.flat_params.cpu().numpy().copy()
    g = gen(4, 32, 64, 4, onehot=False)
    big = []
    for t in range(6):
        img, lab = next(g)
        loss, _ = e.train_step(img, lab, 0.1, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
        big.append(loss)
    ref = e.flat_params.cpu().numpy()
    e.close()
    dp_mean = 0.5 * (np.asarray(out[0][1]) + np.asarray(out[1][1]))
    assert np.abs(dp_mean - np.asarray(big)).max() < 1e-4 * max(1.0, np.abs(big).max()), (dp_mean, big)
    upd_ref, upd_dp = ref - before, out[0][0] - before
    assert np.abs(upd_ref).max() > 0
    assert np.abs(upd_dp - upd_ref).max() <= 5e-3 * np.abs(upd_ref).max(), np.abs(upd_dp - upd_ref).max() / np.abs(upd_ref).max()


def test_onehot_labels_are_converted_on_the_gpu_and_validated():
    from fcn8s_tensorflow_amd.engine import Engine
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = next(gen(2, 32, 64, 4, onehot=False))
    e = Engine(20, widths=SMALL); e.set_params(P)
    l_ids = e.forward_backward(img, lab, keep_prob=1.0)
    for oh in (orc.one_hot(lab, 20), orc.one_hot(lab, 20).astype(np.int32), orc.one_hot(lab, 20).astype(np.float32),
               orc.one_hot(lab, 20).astype(np.uint8)):
        assert e.forward_backward(img, oh, keep_prob=1.0) == l_ids     # same ids -> bitwise the same forward
    e2 = Engine(20, widths=SMALL); e2.set_params(P)
    bad = orc.one_hot(lab, 20).copy(); bad[0, 0, 0, :] = True
    with pytest.raises(ValueError, match="one-hot"):
        e2.forward_backward(img, bad, keep_prob=1.0)
    e.close(); e2.close()


def test_tensorflow_checkpoint_files_round_trip(tmp_path):
    """The reference's on-disk variable format (TF tensor bundle) is read and written without TensorFlow:
    a VGG-16 'SavedModel' directory feeds `vgg16_dir`, an FCN-8s checkpoint feeds `model_load_dir` /
    `load_variables` including the Adam slots and global_step under their TF names."""
    from fcn8s_tensorflow_amd import tf_bundle
    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    m = make()
    g = gen(2, 32, 64, 0, onehot=False)
    m.train(g, epochs=1, steps_per_epoch=2, learning_rate_schedule=lambda s: 1e-3, record_summaries=False)
    params = m.engine.get_params(); mom = m.engine.get_opt_state()
    img = next(g)[0]
    pred = m.predict(img)
    sm_dir = tmp_path / "tf_saved_model"
    m.export_tf_variables(str(sm_dir / "variables" / "variables"))
    names = tf_bundle.read_index(str(sm_dir / "variables" / "variables.index"))[1]
    assert "fc6/weights" in names and "fc6/weights/adam_optimizer_1" in names and "optimizer/global_step" in names
    # encoder-only directory, as downloaded for `vgg16_dir`
    vgg_dir = tmp_path / "vgg16"
    tf_bundle.write_bundle(str(vgg_dir / "variables" / "variables"),
                           {k: v for k, v in params.items() if k.startswith("conv") or k.startswith("fc6/") or k.startswith("fc7/")})
    m.close()

    m2 = FCN8s(model_load_dir=str(sm_dir), tags=["default"])
    assert m2.num_classes == 20 and m2.engine.widths == SMALL and m2.engine.global_step == 2
    for k, v in m2.engine.get_params().items():
        np.testing.assert_array_equal(v, params[k])
    np.testing.assert_array_equal(m2.engine.get_opt_state()[1], mom[1])
    np.testing.assert_array_equal(m2.predict(img), pred)
    m2.close()

    m3 = FCN8s(vgg16_dir=str(vgg_dir), num_classes=20, widths=SMALL)
    got = m3.engine.get_params()
    np.testing.assert_array_equal(got["conv3_2/filter"], params["conv3_2/filter"])
    np.testing.assert_array_equal(got["fc7/biases"], params["fc7/biases"])
    assert np.abs(got["fc7_1x1/kernel"]).max() < 0.01 and m3.engine.global_step == 0     # decoder freshly initialised (:159-160)
    m3.load_variables(str(sm_dir / "variables" / "variables"))                            # tf.train.Saver-style prefix
    np.testing.assert_array_equal(m3.engine.get_params()["fc7_1x1/kernel"], params["fc7_1x1/kernel"])
    assert m3.engine.global_step == 2
    m3.close()


def test_gpu_augmentation_matches_host_crop_flip_brightness():
    """crop / canvas placement, horizontal flip and brightness on the GPU, bit-exact against the host BatchGenerator's
    steps (data_generator/batch_generator.py:268-341, :469-486 through cv2_compat: OpenCV's 8-bit HSV round trip, V scaled in float64,
    saturated, truncated)."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import cv2_compat as cv
    from fcn8s_tensorflow_amd.batch_generator import _place_or_crop
    rng = np.random.default_rng(3)
    N, H, W, Ho, Wo = 4, 20, 28, 16, 24
    img = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8); lab = rng.integers(0, 20, (N, H, W), dtype=np.uint8)
    img[0, :4] = np.linspace(0, 255, 28).astype(np.uint8)[:, None]           # greys (s == 0) and saturated primaries
    img[0, 4, :3] = [[255, 0, 0], [0, 255, 0], [0, 0, 255]]
    offs = np.array([[2, 3], [-3, -5], [4, 0], [0, 0]]); flips = np.array([0, 1, 1, 0]); gains = [0.6180339887, 1.5, None, 1.0]
    e = Engine(20, widths=(8, 8, 8, 8, 8, 16, 16))
    out, lo = e.augment(torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda(), out_hw=(Ho, Wo), offsets=offs, flips=flips, gains=gains,
                        void_class_id=5)
    out, lo = out.cpu().numpy(), lo.cpu().numpy()
    for n in range(N):
        ref = np.zeros((Ho, Wo, 3), np.uint8); rl = np.full((Ho, Wo), 5, np.uint8)
        for y in range(Ho):
            for x in range(Wo):
                sy, sx = y + offs[n, 0], (Wo - 1 - x if flips[n] else x) + offs[n, 1]
                if 0 <= sy < H and 0 <= sx < W:
                    ref[y, x] = img[n, sy, sx]; rl[y, x] = lab[n, sy, sx]
        if gains[n] is not None:
            ref = cv.brightness(ref, gains[n])
        np.testing.assert_array_equal(out[n], ref)
        np.testing.assert_array_equal(lo[n], rl)
    assert not np.array_equal(out[3], img[3, :Ho, :Wo])          # factor 1.0 still runs the 8-bit HSV round trip, as the reference would
    # every RGB value class through the GPU's HSV round trip: a 64 x 64 x 3-plane sweep of the colour cube (262 144 colours in all
    # would be 4 MB; a stride-5 lattice plus the faces covers each sector and both saturations of V)
    v = np.arange(0, 256, 5, dtype=np.uint8)
    cube = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(1, 52, 52 * 52, 3)
    for g in (0.37, 1.0, 1.9):
        got, _ = e.augment(torch.from_numpy(cube).cuda(), None, gains=[g])
        np.testing.assert_array_equal(got[0].cpu().numpy(), cv.brightness(cube[0], g))
    with pytest.raises(ValueError):
        e.augment(torch.zeros(1, 4, 4, 3).cuda())
    e.close()


def test_label_id_export_feeds_the_official_scorer(tmp_path):
    """predict_and_export_label_ids -> label-id PNGs under the Cityscapes file names -> cityscapes_eval.evaluate_directory (the
    restated evalPixelLevelSemanticLabeling file loop): the scores equal those accumulated directly from predict()."""
    from PIL import Image
    from fcn8s_tensorflow_amd import cityscapes_eval as ce
    m = make()
    rng = np.random.default_rng(5)
    names = ["aachen_000000_000019", "aachen_000001_000019", "bonn_000002_000019"]
    direct = ce.PixelLevelEvaluator()
    for nm in names:
        city = nm.split("_")[0]
        (tmp_path / "leftImg8bit" / city).mkdir(parents=True, exist_ok=True); (tmp_path / "gtFine" / city).mkdir(parents=True, exist_ok=True)
        img = rng.integers(0, 256, (32, 64, 3), dtype=np.uint8)
        gt = rng.integers(0, 34, (32, 64)).astype(np.uint8)
        Image.fromarray(img).save(tmp_path / "leftImg8bit" / city / (nm + "_leftImg8bit.png"))
        Image.fromarray(gt).save(tmp_path / "gtFine" / city / (nm + "_gtFine_labelIds.png"))
        direct.add(m.predict([img])[0], gt)
    n = m.predict_and_export_label_ids(str(tmp_path / "results"), str(tmp_path / "leftImg8bit"))
    assert n == 3 and sorted(os.listdir(tmp_path / "results")) == sorted(nm + "_leftImg8bit.png" for nm in names)
    res = ce.evaluate_directory(str(tmp_path / "gtFine" / "*" / "*_gtFine_labelIds.png"), str(tmp_path / "results"))
    np.testing.assert_array_equal(res["confMatrix"], direct.conf)
    assert res["averageScoreClasses"] == direct.results()["averageScoreClasses"] or (np.isnan(res["averageScoreClasses"]) and np.isnan(direct.results()["averageScoreClasses"]))
    import torch
    res_gpu = ce.evaluate_directory(str(tmp_path / "gtFine" / "*" / "*_gtFine_labelIds.png"), str(tmp_path / "results"), device=torch.device("cuda", 0))
    np.testing.assert_array_equal(res_gpu["confMatrix"], direct.conf)
    m.close()


@pytest.mark.parametrize("C", [2, 19])
def test_class_counts_that_are_not_multiples_of_four(C, tmp_path):
    """The reference takes any `num_classes` (e.g. its 2-class KITTI road setup); the library's C-channel tensors are 16-byte
    vectors, so the facade pads the class dimension with classes of probability 0.  Everything the caller sees has C classes and
    equals the oracle evaluated with C classes; the padding subspace stays exactly zero through training."""
    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    m = FCN8s(vgg16_dir='synthetic:1', num_classes=C, widths=SMALL)
    e = m.engine
    assert e.logical_classes == C and e.num_classes % 4 == 0
    P = orc.init_params(C, SMALL, seed=2, decoder_std_scale=30.0, bias_std=0.05)
    e.set_params(P)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 32, 64, 3), dtype=np.uint8)
    lab = rng.integers(0, C, (2, 32, 64), dtype=np.uint8)
    sm = m.predict(img, argmax=False)
    ref = orc.forward(P, img)
    assert sm.shape == (2, 32, 64, C) and np.abs(sm.sum(-1) - 1).max() < 1e-5
    assert np.abs(sm - orc.softmax(ref)).max() < 1e-3
    assert m.predict(img).max() < C
    onehot = orc.one_hot(lab, C)
    loss = e.forward_backward(img, onehot, keep_prob=1.0, l2_rate=1e-3)
    loss_ref, g_ref, _ = orc.loss_and_grads(P, img, onehot.astype(np.float32), l2_rate=1e-3)
    assert abs(loss - loss_ref) < 1e-4 * max(1.0, abs(loss_ref))
    g = e.get_grads()
    for k in g_ref:
        assert g[k].shape == g_ref[k].shape
        assert np.abs(g[k] - g_ref[k]).max() <= 2e-3 * (np.abs(g_ref[k]).max() + 1e-30), k
    # training keeps the padding exactly where it was
    def gen2():
        while True:
            yield img, onehot
    m.train(gen2(), 1, 3, lambda s: 1e-3, metrics={'mean_iou'}, eval_frequency=1)
    pad_w = e.param_view('fc7_1x1/kernel')[..., C:]
    assert float(pad_w.abs().max()) == 0.0 and float(e.param_view('fc7_pool4_pool3_conv2d_trans/bias')[C:].max()) < -1e29
    cm, _, _ = e.metrics_raw()
    assert cm.shape == (C, C) and cm.sum() == 3 * lab.size          # evaluation on the training generator: 3 batches
    m.save(str(tmp_path), 'saved_model', force_save=True)
    d = [x for x in os.listdir(tmp_path) if x.startswith('saved_model')][0]
    m2 = FCN8s(model_load_dir=str(tmp_path / d))
    assert m2.num_classes == C
    np.testing.assert_array_equal(m2.predict(img), m.predict(img))
    m.export_tf_variables(str(tmp_path / 'tfvars'))
    from fcn8s_tensorflow_amd import tf_bundle
    t = tf_bundle.read_bundle(str(tmp_path / 'tfvars'))
    assert t['fc7_conv2d_trans/kernel'].shape == (4, 4, C, C) and t['fc7_1x1/bias/adam_optimizer'].shape == (C,)
    assert 'optimizer/beta1_power' in t and abs(float(np.asarray(t['optimizer/beta1_power']).reshape(-1)[0]) - 0.9 ** (e.global_step + 1)) < 1e-6
    m.close(); m2.close()


def test_import_order_package_before_torch():
    """`from fcn8s_tensorflow_amd.fcn8s import FCN8s` as the first import of a process (the drop-in usage of INTEGRATION.md A) works: the
    package makes sure a single HIP runtime serves torch and libfcn8s_hip.so (with the library's runtime loaded first, fcn8s_create
    failed with 'no ROCm-capable device')."""
    import subprocess, sys
    code = ("from fcn8s_tensorflow_amd.fcn8s import FCN8s\n"
            "import numpy as np\n"
            "m = FCN8s(vgg16_dir='synthetic:0', num_classes=20, widths=(8, 16, 32, 64, 64, 128, 128))\n"
            "p = m.predict(np.zeros((1, 32, 32, 3), np.uint8))\n"
            "assert p.shape == (1, 32, 32)\n"
            "m.close(); print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
