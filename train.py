#!/usr/bin/env python
"""CLI of the reference's train.py (:18-80): same positional / optional arguments.

    python train.py sketch-transformer-tf2 --id exp0 --dataset /data/quickdraw -o /out \\
        --hparams num_layers=4,d_model=128 --base-hparams batch_size=128 --data-hparams token_type=grid
    torchrun --nproc-per-node 8 train.py sketch-transformer-tf2 --data-loader stroke3-synthetic -o /out   # data parallel

``--gpu`` selects the visible device for a single process; under torchrun every rank takes LOCAL_RANK.
"""
import argparse
import os
import pprint


def main():
    parser = argparse.ArgumentParser(description='Train modified transformer with sketch data')
    parser.add_argument("model_name", default=None, help="Model that we are going to train")
    parser.add_argument("--id", default="0", help="experiment signature")
    parser.add_argument("--dataset", default=None, help="Input data folder")
    parser.add_argument("-o", "--output-dir", default="", help="Output directory")
    parser.add_argument("-p", "--hparams", default=None, help="Parameters that are specific to one model")
    parser.add_argument("--base-hparams", default=None, help="Model parameters that concern all models")
    parser.add_argument("--data-hparams", default=None, help="Dataset-related parameters")
    parser.add_argument("-g", "--gpu", default=0, type=int, nargs='+', help="GPU ID to run on")
    parser.add_argument("-r", "--resume", default=None, help="One of 'latest' or a checkpoint name")
    parser.add_argument("--data-loader", default='stroke3-distributed', help="Data loader that will provide data for model")
    parser.add_argument("--help-hps", action="store_true", help="Prints out each hparams default values")
    parser.add_argument("--max-steps", default=None, type=int, help="stop after this many steps (smoke runs)")
    args = parser.parse_args()

    from sketchformer_amd import models, dataloaders, parallel
    from sketchformer_amd.utils import hparams as hp
    Model = models.get_model_by_name(args.model_name)
    DataLoader = dataloaders.get_dataloader_by_name(args.data_loader)
    if args.help_hps:
        print("\nBase model default parameters: \n{}\n\n{} default parameters: \n{}\n\n{} data loader default parameters: \n{}".format(
            pprint.pformat(Model.base_default_hparams().values()), args.model_name,
            pprint.pformat(Model.specific_default_hparams().values()), args.data_loader,
            pprint.pformat(DataLoader.default_hparams().values())))
        return
    model_hps = Model.parse_hparams(base=args.base_hparams, specific=args.hparams)
    data_hps = DataLoader.parse_hparams(args.data_hparams)

    import torch
    rank, world, local_rank, pg = parallel.init_from_env()
    gpu = args.gpu if isinstance(args.gpu, int) else args.gpu[0]
    torch.cuda.set_device(local_rank if world > 1 else gpu)
    if world > 1 and 'seed' in data_hps:
        data_hps.set_hparam('seed', data_hps.seed + rank)
    dataset = DataLoader(data_hps, args.dataset)
    model = Model(model_hps, dataset, args.output_dir, args.id, process_group=pg)
    if args.resume is not None:
        model.restore_checkpoint_if_exists(args.resume)
    if rank == 0:
        hp.save_config(model.config_filepath, hp.combine_hparams_into_one(model_hps, data_hps))
    model.train(max_steps=args.max_steps)


if __name__ == '__main__':
    main()
