#!/usr/bin/env python
"""Cross-patient seq2seq-GRU training on MI355X — counterpart CLI of the reference's
scripts/train_seq2seq.py (same flags: -pt/--patient, -p/--pool_train; same hyper-parameters, fold /
iteration structure and output files), driven by the HIP model, DataModules and Trainer of this package.

    python scripts/train_seq2seq.py -pt S14 -p True
    python -m torch.distributed.run --nproc-per-node 8 scripts/train_seq2seq.py -pt S14 -p True   # data parallel

Extra (not in the reference): --synthetic N runs on seeded synthetic patients when the private
~/data/pt_decoding_data_S62.pkl is not available; --seed fixes folds/augmentations/initialisation; --iters/--folds/--epochs shrink the 50 x 20 x 500 schedule.
"""
import argparse
import csv
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cross_patient_speech_decoding_amd.alignment.alignment_utils as utils                     # noqa: E402
from cross_patient_speech_decoding_amd.alignment.AlignCCA import AlignCCA                       # noqa: E402
from cross_patient_speech_decoding_amd.nn_models import trainer as L                            # noqa: E402
from cross_patient_speech_decoding_amd.nn_models.data_utils.augmentations import (noise_jitter, scaling,   # noqa: E402
                                                                                  time_shifting)
from cross_patient_speech_decoding_amd.nn_models.data_utils.datamodules import (AlignedMicroValDataModule,  # noqa: E402
                                                                                SimpleMicroDataModule)
from cross_patient_speech_decoding_amd.nn_models.models import Seq2SeqRNN                       # noqa: E402


def init_parser():
    parser = argparse.ArgumentParser(description='Seq2seq GRU training (MI355X)')
    parser.add_argument('-pt', '--patient', type=str, required=True, help='Patient ID')
    parser.add_argument('-p', '--pool_train', type=str, default='False', required=False,
                        help='Pool patient data for training')
    parser.add_argument('--synthetic', type=int, default=0, help='use N seeded synthetic patients instead of the pkl')
    parser.add_argument('--iters', type=int, default=50)
    parser.add_argument('--folds', type=int, default=20)
    parser.add_argument('--epochs', type=int, default=500)
    parser.add_argument('--seed', type=int, default=None, help='seed folds/augmentations/initialisation (reference: unseeded)')
    parser.add_argument('--hidden', type=int, default=500)
    parser.add_argument('--out', type=str, default='~/workspace/nn_data')
    return parser


def str2bool(s):
    return s.lower() == 'true'


def load_data(pt, synthetic):
    if synthetic:
        from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
        pats = [make_patient(p, 160 - 6 * p, T=200, C=48 + 8 * p, n_cond=24) for p in range(synthetic)]
        tar = (pats[0][0], pats[0][1][:, 0], pats[0][1])
        pre = [(x, y[:, 0], y) for x, y in pats[1:]]
        return tar, pre
    data_filename = os.path.expanduser('~/data/pt_decoding_data_S62.pkl')
    pt_data = utils.load_pkl(data_filename)
    return utils.decoding_data_from_dict(pt_data, pt, 1, lab_type='phon', algn_type='phon_seq')


def seq2seq_decoding():
    args = init_parser().parse_args()
    pt, pool_train = args.patient, str2bool(args.pool_train)
    if 'RANK' in os.environ and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    rank = dist.get_rank() if dist.is_initialized() else 0
    tar_data, pre_data = load_data(pt, args.synthetic)
    fs = 200
    data = torch.Tensor(tar_data[0])
    align_labels = torch.Tensor(tar_data[2]).long() - 1
    pool_data = [(torch.Tensor(p[0]), torch.Tensor(p[2]).long() - 1, torch.Tensor(p[2]).long() - 1) for p in pre_data]
    augmentations = [time_shifting, noise_jitter, scaling]
    context_prefix = 'pooled' if pool_train else 'ptSpecific'
    batch_size, n_folds, val_size = 5000, args.folds, 0.2
    base = os.path.expanduser(args.out)
    fold_data_path = os.path.join(base, 'datamodules', pt, context_prefix)
    if pool_train:
        dm = AlignedMicroValDataModule(data, align_labels, align_labels, pool_data, AlignCCA, batch_size=batch_size,
                                       process_group=dist.group.WORLD if dist.is_initialized() else None,
                                       folds=n_folds, val_size=val_size, augmentations=augmentations,
                                       data_path=fold_data_path)
    else:
        dm = SimpleMicroDataModule(data, align_labels, batch_size=batch_size, folds=n_folds, val_size=val_size,
                                   augmentations=augmentations, data_path=fold_data_path)
    # model / training parameters (reference :119-147)
    gclip_val, num_classes, n_filters = 0.5, 9, 100
    kernel_size, stride, padding = int(50 * fs / 1000), int(50 * fs / 1000), 0
    n_enc_layers, n_dec_layers, hidden_size = 2, 1, args.hidden
    cnn_dropout = rnn_dropout = 0.3
    learning_rate, l2_reg, activ, model_type = 1e-4, 1e-5, False, 'gru'
    max_epochs = args.epochs
    acc_dir = os.path.join(base, 'accs', pt)
    if rank == 0:
        os.makedirs(os.path.join(acc_dir, context_prefix, 'iters'), exist_ok=True)

    iter_accs = []
    for i in range(args.iters):
        if rank == 0:
            print(f'##### Setting up data module for iteration {i + 1} #####', flush=True)
        if dist.is_initialized() or args.seed is not None:   # identical folds / augmentations on every rank
            L.seed_everything((args.seed if args.seed is not None else 1000) + i)
        dm.setup()
        fold_accs = []
        for fold in range(n_folds):
            dm.set_fold(fold)
            in_channels = dm.get_data_shape()[-1]
            model = Seq2SeqRNN(in_channels, n_filters, hidden_size, num_classes, n_enc_layers, n_dec_layers, kernel_size,
                               stride, padding, cnn_dropout, rnn_dropout, model_type, learning_rate, l2_reg,
                               activation=activ, decay_iters=max_epochs)
            callbacks = [L.ModelCheckpoint(monitor='val_acc', mode='max'), L.LearningRateMonitor(logging_interval='epoch')]
            trainer = L.Trainer(max_epochs=max_epochs, gradient_clip_val=gclip_val, accelerator='auto', callbacks=callbacks,
                                logger=True, enable_progress_bar=False)
            trainer.fit(model, dm.train_dataloader(), dm.val_dataloader())
            trainer.test(model, dm.test_dataloader(), ckpt_path='best')
            fold_accs.append(float(trainer.logged_metrics['test_acc']))
            if rank == 0:
                print(f'iter {i + 1} fold {fold + 1}: test_acc {fold_accs[-1]:.4f}', flush=True)
        iter_accs.append(fold_accs)
        if rank == 0:
            with open(os.path.join(acc_dir, context_prefix, 'iters', f'{pt}_{context_prefix}_iter{i + 1}.csv'), 'w') as f:
                csv.writer(f).writerows(iter_accs)
    if rank == 0:
        np.save(os.path.join(acc_dir, f'{pt}_{context_prefix}_accs.npy'), np.array(iter_accs))
        print('mean test accuracy', float(np.mean(iter_accs)))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    seq2seq_decoding()
