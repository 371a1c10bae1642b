#!/usr/bin/env python3
"""Data-parallel FCN-8s training on one node, the way INTEGRATION.md describes it: one process per GPU, every rank runs the
unchanged `FCN8s.train()` on its own shard of the files, gradients are all-reduced over RCCL inside the step.

    python run_dp.py --gpus 8 --images <leftImg8bit/train> --labels <gtFine/train> [--vgg16 <SavedModel dir>]
    python run_dp.py --gpus 2                      # no dataset: a small synthetic PNG tree is generated

Called without torchrun it launches its own ranks (`python -m torch.distributed.run --nproc-per-node N run_dp.py ...`).
The reference is single-device (fcn8s_tensorflow.py:65); what follows mirrors its tutorial's training cell
(fcn8s_tutorial.ipynb: BatchGenerator -> FCN8s(...) -> model.train(...)) with the imports swapped."""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--images", default=None, help="directory tree of *_leftImg8bit.png (sub-directories = cities)")
    ap.add_argument("--labels", default=None, help="directory tree of *_gtFine_labelIds.png")
    ap.add_argument("--vgg16", default="synthetic:0", help="VGG-16 SavedModel directory, a directory with vgg16_weights.npz, or 'synthetic[:seed]'")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU and step")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--steps-per-epoch", type=int, default=20)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"], help="who all-reduces the gradient buckets: torch.distributed (default) or the "
                    "library's own RCCL communicator behind the C ABI (fcn8s_comm_init / fcn8s_allreduce_bucket); the process group then only carries the id")
    ap.add_argument("--device", type=int, default=None, help="HIP device for every rank (testing on a one-GPU box with --backend gloo); default LOCAL_RANK")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # become the launcher
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))

    import numpy as np
    import torch
    import torch.distributed as dist
    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    from fcn8s_tensorflow_amd.batch_generator import BatchGenerator
    from fcn8s_tensorflow_amd.cityscapes_eval import IDS_TO_TRAINIDS_ARRAY

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if args.device is not None:
        local = args.device
    if world > 1:
        # this rank, the threads of its process group and engine, its feeder thread and the decode workers it forks below: on the CPUs of its
        # GPU's NUMA node -- bound first, so that all of them inherit the mask
        from fcn8s_tensorflow_amd.dp import bind_to_gpu_numa
        numa = bind_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)) if args.device is None else 1)
        if rank == 0:
            print("rank 0 NUMA binding:", numa)
        torch.cuda.set_device(local)
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}))

    images, labels = args.images, args.labels
    if images is None:
        import bench
        root = os.path.join(tempfile.gettempdir(), "fcn8s_run_dp_data")
        if rank == 0 and not os.path.isdir(root):
            bench.make_png_dataset(root, 8 * args.batch, args.height, args.width)
        if world > 1:
            dist.barrier()
        images, labels, convert = os.path.join(root, "images"), os.path.join(root, "gt"), False      # synthetic labels are train ids already
    else:
        convert = IDS_TO_TRAINIDS_ARRAY                                                                 # Cityscapes label ids -> 20 train ids
    data = BatchGenerator(image_dirs=[images], image_file_extension='png', ground_truth_dirs=[labels],
                          image_name_split_separator='_leftImg8bit', ground_truth_suffix='_gtFine_labelIds',
                          check_existence=True, num_classes=20)
    data.shard(rank, world)                                       # every rank its own files
    train_gen = data.generate(batch_size=args.batch, convert_ids_to_ids=convert, convert_to_one_hot=True, void_class_id=0,
                              random_crop=(args.height, args.width), flip=0.5, shuffle=True, workers=args.workers)
    model = FCN8s(vgg16_dir=args.vgg16, num_classes=20, device_id=local)          # weights are broadcast from rank 0
    if args.comm == "native":
        model.engine.comm_init_native()                                           # one RCCL rank per model inside libfcn8s_hip.so (also with a single rank)
        model.engine.broadcast_params(0)
    model.train(train_generator=train_gen, epochs=args.epochs, steps_per_epoch=args.steps_per_epoch,
                learning_rate_schedule=lambda step: 1e-4 if step < 10000 else 1e-5, keep_prob=0.5, l2_regularization=0.0,
                eval_dataset='train', eval_frequency=args.epochs, metrics={'loss', 'mean_iou', 'accuracy'},
                save_during_training=args.save_dir is not None, save_dir=args.save_dir, save_best_only=False, save_frequency=args.epochs,
                record_summaries=False)
    if rank == 0:
        print("rank 0: %d ranks x %d images/step, global step %d, training loss %.4f, metrics %s"
              % (world, args.batch, model.g_step, model.training_loss, dict(zip(model.metric_names, model.metric_values))))
    train_gen.close()
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
