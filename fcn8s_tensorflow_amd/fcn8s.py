"""Drop-in for `fcn8s_tensorflow.FCN8s` (reference: fcn8s_tensorflow.py:17-952).

Same constructor, `train / evaluate / predict / predict_and_save / save /
load_variables / close` signatures, keyword names, defaults, validation
messages and bookkeeping attributes.  The TensorFlow session behind them is
replaced by `engine.Engine` (ctypes -> libfcn8s_hip.so, hand-written gfx950
kernels).  Host-side control flow below restates the reference's loops; the
three `sess.run` sites (:554-572, :685-689, :764-770) become
`Engine.train_step / eval_step / predict`.

Differences a maintainer must know (also listed in INTEGRATION.md):
  * Variables stored by TensorFlow (the VGG-16 SavedModel for `vgg16_dir`,
    FCN-8s SavedModels / Saver checkpoints for `model_load_dir`,
    `variables_load_dir`, `load_variables`) are READ through `tf_bundle.py`
    (tensor-bundle format, no TensorFlow needed; graph files are ignored).
    `save()` keeps the reference's directory naming scheme (:904-920) but
    stores `variables.npz` + `fcn8s_meta.json`; `export_tf_variables()` writes
    a TF-restorable bundle.  `vgg16_dir` may also hold `vgg16_weights.npz` or
    be the string 'synthetic[:seed]'.
  * Labels may also be passed as uint8 class-id maps (N,H,W); one-hot
    (N,H,W,C) input as in the reference is reduced to ids on the GPU.
  * Data-parallel training: launch one process per GPU with torch.distributed
    initialised (RCCL); every rank feeds its own shard of the minibatch.
"""
from __future__ import annotations

import json
import os
import shutil
import sys
import time
import warnings
from collections import deque
from glob import glob

import numpy as np

try:
    from tqdm import trange
except Exception:  # pragma: no cover
    def trange(n, **kw):
        class _R:
            def __init__(self, n): self.n = n
            def __iter__(self): return iter(range(self.n))
            def set_description(self, *a, **k): pass
            def set_postfix(self, *a, **k): pass
        return _R(n)

from . import _lib as L
from . import tf_bundle
from .engine import Engine, padded_classes

_SAVERS = {'saved_model', 'train_saver'}


def _check_metric_names(metrics):
    for metric in metrics:
        if metric not in ('loss', 'mean_iou', 'accuracy'):
            raise ValueError("{} is not a valid metric. Valid metrics are ['loss', mean_iou', 'accuracy']".format(metric))


class FCN8s:

    def __init__(self, model_load_dir=None, tags=None, vgg16_dir=None, num_classes=None, variables_load_dir=None,
                 device_id=None, seed=0, widths=None, fc6_ksize=7):
        '''
        Arguments (first five: fcn8s_tensorflow.py:19-35; the rest are additions):
            model_load_dir (string, optional): directory written by `save(saver='saved_model')`, or a TensorFlow
                SavedModel directory written by the reference (its variables bundle is read).
            tags (list, optional): kept for signature parity; checked against the saved tags if given.
            vgg16_dir (string, optional): the VGG-16 SavedModel directory (variables bundle), a directory with
                `vgg16_weights.npz`, or 'synthetic[:seed]'.
            num_classes (int, optional): number of segmentation classes (any count up to 64; counts that are not a multiple of 4
                are padded inside the engine with classes of probability 0).
            variables_load_dir (string, optional): path prefix written by `save(saver='train_saver')`.
            device_id (int, optional): HIP device; defaults to LOCAL_RANK or 0.
            seed (int): dropout seed (rank is added in data-parallel runs).
        '''
        if (model_load_dir is None) and (vgg16_dir is None or num_classes is None):
            raise ValueError("You must provide either both `model_load_dir` and `tags` or both `vgg16_dir` and `num_classes`.")

        self.variables_load_dir = variables_load_dir
        self.model_load_dir = model_load_dir
        self.tags = tags
        self.vgg16_dir = vgg16_dir
        self.vgg16_tag = 'vgg16'
        self.num_classes = num_classes

        self.variables_updated = False
        self.eval_dataset = None
        # False: mean IoU over the classes that occur (tf.metrics.mean_iou of later TF 1.x); True: over all classes, absent ones
        # counting 0 -- the TF 1.3.0 the tutorial ran on (fcn8s_tutorial.ipynb:311).  Not a reference argument; set it on the instance.
        self.mean_iou_over_all_classes = False

        self.metric_names = []
        self.metric_values = []
        self.best_metric_values = []
        self.metric_value_tensors = []   # kept for attribute parity; hold metric names here
        self.metric_update_ops = []

        self.training_loss = None
        self.best_training_loss = 99999999.9
        self.g_step = None

        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0"))
        rank = int(os.environ.get("RANK", "0"))

        tf_prefix = tf_bundle.find_bundle_prefix(model_load_dir) if model_load_dir is not None else None
        if model_load_dir is not None and tf_prefix is not None and not os.path.isfile(os.path.join(model_load_dir, 'fcn8s_meta.json')):
            # a TensorFlow SavedModel / Saver checkpoint written by the reference (:922-934): variables, Adam slots, global_step
            tensors = tf_bundle.read_bundle(tf_prefix)
            if 'fc7_1x1/bias' not in tensors:
                raise ValueError("'{}' does not hold an FCN-8s checkpoint (no variable 'fc7_1x1/bias').".format(model_load_dir))
            self.num_classes = int(tensors['fc7_1x1/bias'].shape[0])
            w = tuple(int(tensors[k].shape[-1]) for k in ('conv1_2/filter', 'conv2_2/filter', 'conv3_3/filter', 'conv4_3/filter', 'conv5_3/filter', 'fc6/weights', 'fc7/weights'))
            self.engine = Engine(padded_classes(self.num_classes), widths=w, fc6_ksize=int(tensors['fc6/weights'].shape[0]), device_id=device_id, seed=seed + rank,
                                 logical_classes=self.num_classes)
            _load_tf_tensors(self.engine, tensors, with_state=True)
        elif model_load_dir is not None:
            meta = _read_meta(model_load_dir)
            if tags is not None and meta.get("tags") is not None and not set(tags) <= set(meta["tags"]):
                raise RuntimeError("MetaGraphDef associated with tags {} could not be found in SavedModel (available: {}).".format(tags, meta["tags"]))
            self.num_classes = int(meta["num_classes"])
            self.engine = Engine(padded_classes(self.num_classes), widths=meta.get("widths"), fc6_ksize=meta.get("fc6_ksize", 7),
                                 device_id=device_id, seed=seed + rank, logical_classes=self.num_classes)
            _load_checkpoint(self.engine, os.path.join(model_load_dir, "variables", "variables.npz"), with_state=True)
        else:
            self.engine = Engine(padded_classes(num_classes), widths=widths, fc6_ksize=fc6_ksize, device_id=device_id, seed=seed + rank,
                                 logical_classes=num_classes)
            self._load_vgg16()
            if variables_load_dir is not None:
                self.load_variables(variables_load_dir)
        self.engine.broadcast_params(0)
        self.engine.metrics_reset()

    # fcn8s_tensorflow.py:127-152 -- the encoder weights enter here
    def _load_vgg16(self):
        d = str(self.vgg16_dir)
        if d.startswith('synthetic'):
            s = int(d.split(':', 1)[1]) if ':' in d else 0
            self.engine.init_params(seed=s)
            return
        prefix = tf_bundle.find_bundle_prefix(d)
        if prefix is not None:                       # the reference's VGG-16 SavedModel (README.md:42): variables bundle
            self.engine.init_params(seed=0)          # decoder: truncated normal (:159-160); encoder overwritten below
            tensors = tf_bundle.read_bundle(prefix)
            params = {k: v for k, v in tensors.items() if k in self.engine.specs and (k.startswith('conv') or k.startswith('fc6/') or k.startswith('fc7/'))}
            missing = [k for k in self.engine.specs if (k.startswith('conv') or k.startswith('fc6/') or k.startswith('fc7/')) and k not in params]
            if missing:
                raise ValueError("the VGG-16 checkpoint lacks variables: {}".format(missing[:5]))
            self.engine.set_params(params)
            return
        npz = os.path.join(d, 'vgg16_weights.npz')
        if os.path.isfile(npz):
            self.engine.init_params(seed=0)          # decoder: truncated normal (:159-160); encoder overwritten below
            data = np.load(npz)
            params = {k: data[k] for k in data.files if k in self.engine.specs}
            missing = [k for k in self.engine.specs if ('conv' in k.split('/')[0] and 'trans' not in k or k.startswith('fc6/') or k.startswith('fc7/')) and k not in params]
            if missing:
                raise ValueError("vgg16_weights.npz lacks variables: {}".format(missing[:5]))
            self.engine.set_params(params)
            return
        raise ValueError("`vgg16_dir` must contain a TensorFlow SavedModel (variables/variables.index), vgg16_weights.npz, or be 'synthetic[:seed]', got '{}'.".format(d))

    def _initialize_metrics(self, metrics):
        '''fcn8s_tensorflow.py:371-397'''
        table = (('loss', 99999999.9, 'mean_loss'), ('mean_iou', 0.0, 'mean_iou'), ('accuracy', 0.0, 'acc'))
        chosen = [row for row in table if row[0] in metrics]
        self.metric_names = [name for name, _, _ in chosen]
        self.best_metric_values = [worst for _, worst, _ in chosen]
        self.metric_update_ops = [stem + '_update_op' for _, _, stem in chosen]      # names only: there is no graph
        self.metric_value_tensors = [stem + '_value' for _, _, stem in chosen]

    def train(self,
              train_generator,
              epochs,
              steps_per_epoch,
              learning_rate_schedule,
              keep_prob=0.5,
              l2_regularization=0.0,
              eval_dataset='train',
              eval_frequency=5,
              val_generator=None,
              val_steps=None,
              metrics={},
              save_during_training=False,
              save_dir=None,
              save_best_only=True,
              save_tags=['default'],
              save_name='',
              save_frequency=5,
              saver='saved_model',
              monitor='loss',
              record_summaries=True,
              summaries_frequency=10,
              summaries_dir=None,
              summaries_name=None,
              training_loss_display_averaging=3):
        '''Trains the model; arguments as fcn8s_tensorflow.py:424-503.  Summaries are written as TensorBoard
        event files (`<summaries_dir>/<summaries_name>[_eval]/events.out.tfevents.*`: total_loss, learning_rate and
        mean / stddev / max / min / histogram of the ten watched weight-bias pairs, :331-366) and, for reading
        without TensorBoard, as JSON lines (`scalars.jsonl`) next to them; if `summaries_dir` is None nothing is recorded.'''
        _check_metric_names(metrics)
        if eval_dataset not in ('train', 'val'):
            raise ValueError("`eval_dataset` must be one of 'train' or 'val', but is '{}'.".format(eval_dataset))
        if eval_dataset == 'val' and (val_generator is None or val_steps is None):
            raise ValueError("When eval_dataset == 'val', a `val_generator` and `val_steps` must be passed.")
        if monitor != 'loss' and monitor not in metrics:
            raise ValueError('You are trying to monitor {}, but it is not in `metrics` and is therefore not being computed.'.format(monitor))

        self.eval_dataset = eval_dataset
        self._initialize_metrics(metrics)
        self.g_step = self.engine.global_step
        self._lr = learning_rate_schedule(self.g_step)       # the schedule is called once here and once after every step (:527, :583)

        train_log = eval_log = None
        if record_summaries and summaries_dir is not None and self.engine.rank == 0:
            run = summaries_name or 'training'
            train_log = _ScalarLog(os.path.join(summaries_dir, run), engine=self.engine)
            if metrics:
                eval_log = _ScalarLog(os.path.join(summaries_dir, run + '_eval'))

        eval_source = {'train': (train_generator, steps_per_epoch, 'Evaluation on training dataset'),
                       'val': (val_generator, val_steps, 'Evaluation on validation dataset')}[eval_dataset]

        try:
            for epoch in range(1, epochs + 1):
                self._run_epoch(train_generator, steps_per_epoch, learning_rate_schedule, keep_prob, l2_regularization,
                                'Epoch {}/{}'.format(epoch, epochs), training_loss_display_averaging,
                                train_log, summaries_frequency)
                eval_epoch = epoch % eval_frequency == 0

                if metrics and eval_epoch:
                    generator, num_batches, description = eval_source
                    self._evaluate(generator, metrics, num_batches, l2_regularization, description)
                    if eval_log is not None:
                        eval_log.add(self.g_step, **{('mean_loss' if n == 'loss' else n): v for n, v in zip(self.metric_names, self.metric_values)})    # tags of :360-362

                if save_during_training and epoch % save_frequency == 0 and self._wants_save(save_best_only, monitor):
                    self.save(model_save_dir=save_dir, saver=saver, tags=save_tags, name=save_name,
                              include_global_step=True, include_last_training_loss=True,
                              include_metrics=bool(self.metric_names))

                # Bests are updated after the save decision (fcn8s_tensorflow.py:648-658).
                self.best_training_loss = min(self.best_training_loss, self.training_loss)
                if eval_epoch:
                    for i, name in enumerate(self.metric_names):
                        if self._improved(name, i):
                            self.best_metric_values[i] = self.metric_values[i]
        finally:
            for log in (train_log, eval_log):          # the event files are complete and closed when train() returns or raises
                if log is not None:
                    log.close()

    def _run_epoch(self, generator, steps, schedule, keep_prob, l2_rate, title, window, log, log_every):
        '''One epoch of train steps (fcn8s_tensorflow.py:542-590): a step runs with the schedule's value
        at the global step before it; `training_loss` is the mean over the last `window` steps.'''
        recent = deque(maxlen=window)
        bar = trange(steps, file=sys.stdout, disable=self.engine.rank != 0)
        bar.set_description(title)
        feed = _Feeder(self.engine, generator, steps)        # batch k+1 is decoded / staged / copied while step k runs
        try:
            for _ in bar:
                lr = self._lr
                images, labels = feed.next()
                loss, self.g_step = self.engine.train_step(images, labels, learning_rate=lr, keep_prob=keep_prob, l2_rate=l2_rate)
                self.variables_updated = True
                if log is not None and (self.g_step - 1) % log_every == 0:
                    log.add(self.g_step, total_loss=loss, learning_rate=lr)
                recent.append(loss)
                self.training_loss = float(np.mean(recent))
                bar.set_postfix(ordered_dict={'loss': self.training_loss, 'learning rate': lr})
                self._lr = schedule(self.g_step)
        finally:
            feed.close()

    def _improved(self, name, i):
        '''Whether metric `i` beat its best.  The reference compares the name against
        ['accuracry', 'mean_iou'] (fcn8s_tensorflow.py:626,:656), so 'accuracy' never counts as an
        improvement there; kept, because `best_metric_values` and save decisions depend on it.'''
        if name == 'loss':
            return self.metric_values[i] < self.best_metric_values[i]
        if name == 'mean_iou':
            return self.metric_values[i] > self.best_metric_values[i]
        return False

    def _wants_save(self, best_only, monitor):
        '''Save decision of fcn8s_tensorflow.py:612-634.'''
        if not best_only:
            return True
        if monitor == 'loss' and 'loss' not in self.metric_names:
            better = self.training_loss < self.best_training_loss
        else:
            better = self._improved(monitor, self.metric_names.index(monitor))
        if better:
            print('New best {} value, saving model.'.format(monitor))
        else:
            print('No improvement over previous best {} value, not saving model.'.format(monitor))
        return better

    def _evaluate(self, data_generator, metrics, num_batches, l2_regularization, description='Running evaluation'):
        '''fcn8s_tensorflow.py:660-697'''
        self.engine.metrics_reset()

        tr = trange(num_batches, file=sys.stdout, disable=self.engine.rank != 0)
        tr.set_description(description)

        self.engine.freeze(True)            # no training inside an evaluation loop: transformed filters are built once
        feed = _Feeder(self.engine, data_generator, num_batches)
        try:
            for step in tr:
                batch_images, batch_labels = feed.next()
                self.engine.eval_step(batch_images, batch_labels, l2_rate=l2_regularization)
        finally:
            feed.close()
            self.engine.freeze(False)

        self.engine.metrics_allreduce()
        values = dict(zip(('loss', 'mean_iou', 'accuracy'), self.engine.metrics_get(all_classes=self.mean_iou_over_all_classes)))
        self.metric_values = [values[n] for n in self.metric_names]

        if self.engine.rank == 0:
            print(''.join('{}: {:.4f}  '.format(n, v) for n, v in zip(self.metric_names, self.metric_values)))

    def evaluate(self, data_generator, num_batches, metrics={'loss', 'mean_iou', 'accuracy'}, l2_regularization=0.0, dataset='val'):
        '''fcn8s_tensorflow.py:699-741'''
        _check_metric_names(metrics)
        if dataset not in ('train', 'val'):
            raise ValueError("`dataset` must be either 'train' or 'val'.")
        self._initialize_metrics(metrics)
        self._evaluate(data_generator, metrics, num_batches, l2_regularization, description='Running evaluation')
        self.eval_dataset = dataset

    def predict(self, images, argmax=True):
        '''fcn8s_tensorflow.py:743-770.  `images`: array-like of rank 4 (a list of HWC arrays works).
        Returns int64 class ids (N,H,W) or the float32 softmax (N,H,W,C).'''
        if isinstance(images, (list, tuple)):
            images = np.asarray(images)
        return self.engine.predict(images, argmax=argmax)

    def predict_and_save(self,
                         results_dir,
                         images_dir,
                         color_map,
                         resize=False,
                         image_file_extension='png',
                         include_unprocessed_image=False,
                         arrangement='vertical',
                         overwrite_existing=True):
        '''fcn8s_tensorflow.py:772-855 (PIL instead of scipy.misc / helpers.visualization_utils).'''
        from PIL import Image

        if overwrite_existing and os.path.exists(results_dir):
            shutil.rmtree(results_dir)
        os.makedirs(results_dir)

        image_paths = glob(os.path.join(images_dir, '*.' + image_file_extension))
        num_images = len(image_paths)

        print('The segmented images will be saved to "{}"'.format(results_dir))

        tr = trange(num_images, file=sys.stdout)
        tr.set_description('Processing images')

        self.engine.freeze(True)            # constant weights for the whole directory
        try:
            for i in tr:
                self._segment_file(image_paths[i], results_dir, color_map, resize, include_unprocessed_image, arrangement)
        finally:
            self.engine.freeze(False)

    def predict_and_export_label_ids(self, results_dir, images_dir, resize=False, image_file_extension='png', overwrite_existing=True):
        '''Not in the reference: runs every `*.png` below `images_dir` (sub-directories = cities, as in leftImg8bit/val) through the
        model and writes the argmax as single-channel label-id PNGs under the same file names into `results_dir` -- the input of the
        official scorer (cityscapesscripts/evaluation/evalPixelLevelSemanticLabeling.py:72-106, 553-555; cityscapes_eval.evaluate_directory
        here).  Predictions are train ids 0..19 (0 = void) mapped through labels.py:188-192; `resize=(h, w)` feeds the network a
        resized image and writes the prediction back at the file's own size (nearest neighbour).'''
        from PIL import Image
        from . import cityscapes_eval as ce
        if self.num_classes != 20:
            raise ValueError("label-id export uses the 20 Cityscapes train ids; this model has {} classes.".format(self.num_classes))
        if overwrite_existing and os.path.exists(results_dir):
            shutil.rmtree(results_dir)
        os.makedirs(results_dir, exist_ok=True)
        paths = sorted(glob(os.path.join(images_dir, '**', '*.' + image_file_extension), recursive=True))
        tr = trange(len(paths), file=sys.stdout)
        tr.set_description('Exporting label ids')
        self.engine.freeze(True)
        try:
            for i in tr:
                pil = Image.open(paths[i]).convert('RGB')
                size = pil.size
                if resize and not np.array_equal((pil.height, pil.width), resize):
                    pil = pil.resize((resize[1], resize[0]), Image.BILINEAR)
                pred = np.asarray(self.predict([np.asarray(pil)], argmax=True))[0]
                ids = Image.fromarray(ce.TRAINIDS_TO_IDS_ARRAY[pred])
                if ids.size != size:
                    ids = ids.resize(size, Image.NEAREST)
                ids.save(os.path.join(results_dir, os.path.basename(paths[i])))
        finally:
            self.engine.freeze(False)
        return len(paths)

    def _segment_file(self, filepath, results_dir, color_map, resize, include_unprocessed_image, arrangement):
        '''Loop body of predict_and_save (fcn8s_tensorflow.py:829-855).'''
        from PIL import Image
        pil = Image.open(filepath).convert('RGB')
        if resize and not np.array_equal((pil.height, pil.width), resize):
            pil = pil.resize((resize[1], resize[0]), Image.BILINEAR)
        image = np.asarray(pil)
        img_height, img_width, img_ch = image.shape

        prediction = self.predict([image], argmax=False)
        processed = print_segmentation_onto_image(image=image, prediction=prediction, color_map=color_map)

        if include_unprocessed_image:
            if arrangement == 'vertical':
                canvas = Image.new('RGB', (img_width, 2 * img_height))
                canvas.paste(processed, (0, 0)); canvas.paste(pil, (0, img_height))
            else:
                canvas = Image.new('RGB', (2 * img_width, img_height))
                canvas.paste(processed, (0, 0)); canvas.paste(pil, (img_width, 0))
            processed = canvas

        processed.save(os.path.join(results_dir, os.path.basename(filepath)))

    def save(self,
             model_save_dir,
             saver,
             tags=['default'],
             name=None,
             include_global_step=True,
             include_last_training_loss=True,
             include_metrics=True,
             force_save=False):
        '''fcn8s_tensorflow.py:857-936: same guard, same validation, same directory name.'''
        if (not self.variables_updated) and (not force_save):
            print("Abort: Nothing to save, no training has been performed since the model was last saved.")
            return

        if not saver in _SAVERS:
            raise ValueError("Unexpected value for `saver`: Can be either 'saved_model' or 'train_saver', but received '{}'.".format(saver))

        if self.training_loss is None:
            include_last_training_loss = False

        model_name = 'saved_model'
        if not name is None:
            model_name += '_' + name
        if include_global_step:
            self.g_step = self.engine.global_step
            model_name += '_(globalstep-{})'.format(self.g_step)
        if include_last_training_loss:
            model_name += '_(trainloss-{:.4f})'.format(self.training_loss)
        if include_metrics:
            if self.eval_dataset == 'val':
                model_name += '_(eval_on_val_dataset)'
            else:
                model_name += '_(eval_on_train_dataset)'
            for i in range(len(self.metric_names)):
                model_name += '_({}-{:.4f})'.format(self.metric_names[i], self.metric_values[i])
        if not (include_global_step or include_last_training_loss or include_metrics) and (name is None):
            model_name += '_{}'.format(time.time())

        if self.engine.rank == 0:
            target = os.path.join(model_save_dir, model_name)
            if saver == 'saved_model':
                if os.path.exists(target):
                    raise AssertionError("Export directory already exists. Please specify a different export directory: {}".format(target))
                os.makedirs(os.path.join(target, 'variables'))
                _save_checkpoint(self.engine, os.path.join(target, 'variables', 'variables.npz'))
                _write_meta(target, self.engine, tags)
            else:
                os.makedirs(target, exist_ok=True)
                _save_checkpoint(self.engine, os.path.join(target, 'variables.npz'))
                _write_meta(target, self.engine, tags)      # (the reference builds a fresh Saver per save() call, :927-934, so its
                                                            #  max_to_keep=5 never deletes an earlier checkpoint: nothing is pruned here either)
        self.last_saved_model_name = model_name

        self.variables_updated = False

    def load_variables(self, path):
        '''fcn8s_tensorflow.py:938-944.  `path` is the prefix `<dir>/<model_name>/variables`
        (as passed to tf.train.Saver.restore) or the `.npz` file itself.'''
        if os.path.isfile(path + '.index'):          # a tf.train.Saver checkpoint written by the reference
            _load_tf_tensors(self.engine, tf_bundle.read_bundle(path), with_state=True)
            return
        f = path if path.endswith('.npz') else path + '.npz'
        if not os.path.isfile(f):
            raise ValueError("The passed save_path is not a valid checkpoint: {}".format(path))
        _load_checkpoint(self.engine, f, with_state=True)

    def export_tf_variables(self, prefix, include_optimizer_state=True):
        '''Not in the reference: writes all variables (and the Adam slots / global_step under the names
        tf.train.AdamOptimizer(name='adam_optimizer') gives them) as a TensorFlow tensor bundle
        `<prefix>.index` + `<prefix>.data-00000-of-00001`, restorable with tf.train.Saver.'''
        tensors = dict(self.engine.get_params())
        if include_optimizer_state:
            m, v = self.engine.get_opt_state()
            for k, (shape, off) in self.engine.specs.items():
                n = int(np.prod(shape))
                tensors[k + tf_bundle.ADAM_M_SUFFIX] = self.engine.unpad(k, m[off:off + n].reshape(shape))
                tensors[k + tf_bundle.ADAM_V_SUFFIX] = self.engine.unpad(k, v[off:off + n].reshape(shape))
            step = self.engine.global_step
            tensors['optimizer/global_step'] = np.asarray(step, dtype=np.int32)
            # AdamOptimizer's non-slot variables (initialised to beta, multiplied by beta in every apply): a Saver over all
            # saveable variables (load_variables, :943) would not restore without them
            tensors['optimizer/beta1_power'] = np.asarray(0.9 ** (step + 1), dtype=np.float32)
            tensors['optimizer/beta2_power'] = np.asarray(0.999 ** (step + 1), dtype=np.float32)
        tf_bundle.write_bundle(prefix, tensors)

    def close(self):
        '''fcn8s_tensorflow.py:946-952'''
        self.engine.close()
        print("The session has been closed.")


# ---------------------------------------------------------------------------------------------
# checkpoint format: variables by reference name + Adam slots + global_step
# ---------------------------------------------------------------------------------------------
def _save_checkpoint(engine, path):
    arrays = dict(engine.get_params())
    m, v = engine.get_opt_state()
    arrays['__adam_m__'] = m
    arrays['__adam_v__'] = v
    arrays['optimizer/global_step'] = np.asarray(engine.global_step, dtype=np.int64)
    tmp = path + '.tmp.npz'
    np.savez(tmp, **arrays)
    os.replace(tmp, path)


def _load_checkpoint(engine, path, with_state=True):
    data = np.load(path)
    engine.set_params({k: data[k] for k in data.files if k in engine.specs})
    if with_state:
        if '__adam_m__' in data.files:
            engine.set_opt_state(data['__adam_m__'], data['__adam_v__'])
        if 'optimizer/global_step' in data.files:
            engine.global_step = int(data['optimizer/global_step'])


def _load_tf_tensors(engine, tensors, with_state=True):
    engine.set_params({k: v for k, v in tensors.items() if k in engine.specs})
    if not with_state:
        return
    n = engine.flat_params.numel()
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32); found = False
    for k, (shape, off) in engine.specs.items():
        cnt = int(np.prod(shape))
        if k + tf_bundle.ADAM_M_SUFFIX in tensors and k + tf_bundle.ADAM_V_SUFFIX in tensors:
            m[off:off + cnt] = engine.pad(k, tensors[k + tf_bundle.ADAM_M_SUFFIX], slot=True).reshape(-1)
            v[off:off + cnt] = engine.pad(k, tensors[k + tf_bundle.ADAM_V_SUFFIX], slot=True).reshape(-1)
            found = True
    if found:
        engine.set_opt_state(m, v)
    for key in ('optimizer/global_step', 'global_step'):
        if key in tensors:
            engine.global_step = int(np.asarray(tensors[key]).reshape(-1)[0])
            break


def _write_meta(target, engine, tags):
    with open(os.path.join(target, 'fcn8s_meta.json'), 'w') as f:
        json.dump({'format': 'fcn8s_tensorflow_amd/1', 'num_classes': engine.logical_classes, 'widths': list(engine.widths),
                   'fc6_ksize': engine.specs['fc6/weights'][0][0], 'tags': list(tags) if tags else None,
                   'global_step': engine.global_step}, f)


def _read_meta(model_dir):
    f = os.path.join(model_dir, 'fcn8s_meta.json')
    if not os.path.isfile(f):
        if os.path.isfile(os.path.join(model_dir, 'saved_model.pb')):
            raise NotImplementedError("'{}' is a TensorFlow SavedModel; it cannot be loaded without TensorFlow.".format(model_dir))
        raise IOError("SavedModel file does not exist at: {}".format(model_dir))
    with open(f) as fh:
        return json.load(fh)


class _Feeder:
    """Pulls exactly `count` batches from a generator on a helper thread, one batch ahead of the consumer, and -- when the batch is
    (uint8 images, uint8 class ids) -- already starts its host-to-device copy through a staging slot (Engine.stage), so that PNG
    decoding, the pinned-memory copy and the DMA of batch k+1 overlap the GPU work of step k.  The reference pulls and feeds
    serially inside the step loop (fcn8s_tensorflow.py:551-572).  Generators made by this package's BatchGenerator offer
    `next_ids()` (the same batch with class ids instead of the 20x larger one-hot rows); any other generator is consumed through
    next() and its arrays are fed as they are.  Exactly `count` batches are consumed, as in the reference's loops."""

    def __init__(self, engine, generator, count):
        import queue
        import threading
        self.q = queue.Queue(maxsize=1)
        self.err = None
        self.stop = threading.Event()

        def put(item):
            while not self.stop.is_set():
                try:
                    self.q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                pull = getattr(generator, 'next_ids', None)
                for i in range(count):
                    if self.stop.is_set():
                        return
                    images, labels = pull() if pull is not None else next(generator)
                    if (isinstance(images, np.ndarray) and images.dtype == np.uint8 and isinstance(labels, np.ndarray)
                            and labels.dtype == np.uint8 and labels.ndim == 3 and images.ndim == 4):
                        item = (engine.stage(images, labels, slot=i % L.NUM_STAGE_SLOTS), None)
                    else:
                        item = (images, labels)
                    if not put(item):
                        return
            except BaseException as e:          # surfaces in the consumer
                self.err = e
                put(None)

        self.t = threading.Thread(target=work, daemon=True)
        self.t.start()

    def next(self):
        item = self.q.get()
        if item is None:
            raise self.err
        return item

    def close(self):
        """Stops the helper thread and waits for it: after a step raised (or Ctrl-C) it must not keep pulling batches from the
        user's generator, nor sit inside fcn8s_stage_inputs while the model is being closed."""
        import queue
        self.stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.t.join()


class _ScalarLog:
    """One of the reference's two tf.summary.FileWriters (fcn8s_tensorflow.py:531-535): a TensorBoard event file
    (`events.out.tfevents.*`, tf_events.py) in `logdir`, plus the same scalars as JSON lines (`scalars.jsonl`) for reading without
    TensorBoard.  With an engine, every record also carries what `_build_summary_ops` (:331-350) attaches to the ten watched
    weight / bias pairs: mean, stddev, max, min and a histogram (helpers/tf_variable_summaries.py:3-20), computed on the GPU."""

    def __init__(self, logdir, engine=None):
        from .tf_events import EventFileWriter
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'scalars.jsonl')
        self.events = EventFileWriter(logdir)
        self.engine = engine

    def add(self, step, **scalars):
        from . import tf_events
        with open(self.path, 'a') as f:
            f.write(json.dumps(dict(step=int(step), **{k: float(v) for k, v in scalars.items()})) + '\n')
        summary = b''
        if self.engine is not None:
            for name, scope in tf_events.WATCHED_VARIABLES:
                if name in self.engine.specs:
                    summary += tf_events.add_variable_summaries(self.engine.logical_param_view(name), scope)
        summary += b''.join(tf_events.scalar_value(k, v) for k, v in scalars.items())
        self.events.add_summary(summary, int(step))

    def close(self):
        self.events.close()


def create_video_from_images(video_output_name, image_input_dir, frame_rate=30.0, image_file_extension='png'):
    '''helpers/visualization_utils.py:102-120: a video from the images of a directory (sorted by name).  The reference encodes
    MP4 through moviepy (ffmpeg); when moviepy is importable that is what happens here, otherwise -- this image has neither moviepy
    nor ffmpeg -- the frames are written as Motion-JPEG into an AVI container (`<video_output_name>.avi`, plays everywhere, no
    dependencies beyond Pillow).  Returns the path written.'''
    image_paths = sorted(glob(os.path.join(image_input_dir, '*.' + image_file_extension)))
    if not image_paths:
        raise ValueError("no '*.{}' images in {}".format(image_file_extension, image_input_dir))
    try:
        from moviepy.editor import ImageSequenceClip
    except ImportError:
        ImageSequenceClip = None
    if ImageSequenceClip is not None:
        out = "{}.mp4".format(video_output_name)
        ImageSequenceClip(image_paths, fps=frame_rate).write_videofile(out)
        return out
    return _write_mjpeg_avi("{}.avi".format(video_output_name), image_paths, frame_rate)


def _write_mjpeg_avi(path, image_paths, frame_rate, quality=90):
    '''RIFF/AVI 1.0 with one MJPG video stream: hdrl (avih + strl(strh, strf)), movi (one '00dc' chunk per JPEG frame), idx1.'''
    import io
    import struct
    from PIL import Image
    frames, size = [], None
    for p in image_paths:
        im = Image.open(p).convert('RGB')
        if size is None:
            size = im.size
        elif im.size != size:
            im = im.resize(size)
        buf = io.BytesIO()
        im.save(buf, format='JPEG', quality=quality)
        frames.append(buf.getvalue())
    w, h = size
    n = len(frames)
    usec = int(round(1e6 / float(frame_rate)))
    rate, scale = int(round(float(frame_rate) * 1000)), 1000
    maxb = max(len(f) for f in frames)

    def chunk(fourcc, data):
        return fourcc + struct.pack('<I', len(data)) + data + (b'\x00' if len(data) & 1 else b'')

    def lst(kind, data):
        return b'LIST' + struct.pack('<I', 4 + len(data)) + kind + data

    avih = struct.pack('<14I', usec, maxb * rate // scale, 0, 0x10, n, 0, 1, maxb, w, h, 0, 0, 0, 0)           # 0x10 = AVIF_HASINDEX
    strh = b'vids' + b'MJPG' + struct.pack('<IHHIIIIIIII', 0, 0, 0, 0, scale, rate, 0, n, maxb, 0xFFFFFFFF, 0) + struct.pack('<4H', 0, 0, w, h)
    strf = struct.pack('<IiiHH4sIiiII', 40, w, h, 1, 24, b'MJPG', w * h * 3, 0, 0, 0, 0)
    hdrl = lst(b'hdrl', chunk(b'avih', avih) + lst(b'strl', chunk(b'strh', strh) + chunk(b'strf', strf)))
    movi_data, index, off = b'', b'', 4
    for f in frames:
        c = chunk(b'00dc', f)
        index += b'00dc' + struct.pack('<III', 0x10, off, len(f))           # 0x10 = AVIIF_KEYFRAME; offset relative to 'movi'
        movi_data += c
        off += len(c)
    body = b'AVI ' + hdrl + lst(b'movi', movi_data) + chunk(b'idx1', index)
    with open(path, 'wb') as fh:
        fh.write(b'RIFF' + struct.pack('<I', len(body)) + body)
    return path


def print_segmentation_onto_image(image, prediction, color_map):
    '''helpers/visualization_utils.py:7-52: RGBA overlay of the argmax map (returns a PIL image).'''
    from PIL import Image
    if (image.shape[0] != prediction.shape[1]) or (image.shape[1] != prediction.shape[2]):
        raise ValueError("'image' and 'prediction' must have the same height and width, but image has spatial dimensions ({}, {}) and prediction has spatial dimensions ({}, {}).".format(image.shape[0], image.shape[1], prediction.shape[1], prediction.shape[2]))
    mask = np.zeros(shape=(image.shape[0], image.shape[1], 4), dtype=np.uint8)
    segmentation_map = np.squeeze(np.argmax(prediction, axis=-1))
    for segmentation_class, color_value in color_map.items():
        mask[segmentation_map == segmentation_class] = color_value
    mask = Image.fromarray(mask, mode='RGBA')
    out = Image.fromarray(np.asarray(image, dtype=np.uint8)).convert('RGB')
    out.paste(mask, box=None, mask=mask)
    return out
