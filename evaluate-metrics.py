#!/usr/bin/env python
"""CLI of the reference's evaluate-metrics.py: rebuild a trained model from its config.json, restore a checkpoint,
compute the chosen slow metrics, save their plot.

    python evaluate-metrics.py sketch-transformer-tf2 --id exp0 -o /out --dataset /data --metrics val-clas-acc tsne
"""
import argparse
import json
import pprint


def main():
    parser = argparse.ArgumentParser(description='Evaluate a trained sketch transformer')
    parser.add_argument("model_name", default=None, help="Model that we are going to evaluate")
    parser.add_argument("--id", default="0", help="experiment signature")
    parser.add_argument("--data-loader", default='stroke3-distributed', help="Data loader that will provide data for model")
    parser.add_argument("--dataset", default=None, help="Input data folder")
    parser.add_argument("-o", "--output-dir", default="", help="output directory")
    parser.add_argument('-p', "--hparams", default=None, help="Parameters to override")
    parser.add_argument("-g", "--gpu", default=0, type=int, nargs='+', help="GPU ID to run on")
    parser.add_argument('--metrics', type=str, nargs='+', help="selection of metrics you want to calculate")
    parser.add_argument("--help-hps", action="store_true", help="Prints out the hparams file")
    parser.add_argument("-r", "--resume", default='latest', help="One of 'latest' or a checkpoint name")
    args = parser.parse_args()

    from sketchformer_amd import dataloaders, metrics, models
    from sketchformer_amd.utils import hparams as hp
    Model = models.get_model_by_name(args.model_name)
    DataLoader = dataloaders.get_dataloader_by_name(args.data_loader)
    hps = hp.combine_hparams_into_one(Model.default_hparams(), DataLoader.default_hparams())
    hp.load_config(hps, Model.get_config_filepath(args.output_dir, args.id))
    if args.help_hps:
        print("\nLoaded parameters: \n{}\n\n".format(pprint.pformat(hps.values())))
        return
    if args.hparams:
        hps.parse(args.hparams)
    import torch
    torch.cuda.set_device(args.gpu if isinstance(args.gpu, int) else args.gpu[0])
    dataset = DataLoader(hps, args.dataset)
    model = Model(hps, dataset, args.output_dir, args.id)
    model.restore_checkpoint_if_exists(args.resume)
    metrics_list = {m: metrics.build_metric_by_name(m, hps.values()) for m in (args.metrics or [])}
    model.compute_metrics_from(metrics_list)
    plot = model.plot_and_send_notification_for(metrics_list)
    print(json.dumps({m: v.last_value_repr for m, v in metrics_list.items()}))
    if plot:
        print("plots:", plot)
    model.clean_up_tmp_dir()


if __name__ == '__main__':
    main()
