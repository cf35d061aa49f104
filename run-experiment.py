#!/usr/bin/env python
"""CLI of the reference's run-experiment.py.

    python run-experiment.py extract-embeddings --model-name sketch-transformer-tf2 --model-id exp0 -o /out --dataset /data
"""
import argparse
import pprint


def main():
    parser = argparse.ArgumentParser(description='Run an experiment on a trained sketch transformer')
    parser.add_argument("experiment_name", default=None, help="Reference name of experiment that you want to run")
    parser.add_argument("--id", default="0", help="Experiment signature")
    parser.add_argument("-o", "--output-dir", default="", help="output directory")
    parser.add_argument("--exp-hparams", default=None, help="Parameters to override defaults for experiment")
    parser.add_argument("--model-hparams", default=None, help="Parameters to override defaults for model")
    parser.add_argument("-g", "--gpu", default=0, type=int, nargs='+', help="GPU ID to run on")
    parser.add_argument("--model-name", default=None, help="Model that you want to experiment on")
    parser.add_argument("--model-id", default=None, help="Id of the model that you want to experiment on")
    parser.add_argument("--data-loader", default='stroke3-distributed', help="Data loader that will provide data for model")
    parser.add_argument("--dataset", default=None, help="Input data folder if you want to load a model")
    parser.add_argument("-r", "--resume", default='latest', help="One of 'latest' or a checkpoint name")
    parser.add_argument("--help-hps", action="store_true", help="Prints out the hparams default values")
    args = parser.parse_args()

    from sketchformer_amd import dataloaders, experiments, models
    from sketchformer_amd.utils import hparams as hp
    Experiment = experiments.get_experiment_by_name(args.experiment_name)
    if args.help_hps:
        print("\nDefault params for experiment {}: \n{}\n\n".format(
            args.experiment_name, pprint.pformat(Experiment.default_hparams().values())))
        return
    model = None
    if Experiment.requires_model:
        import torch
        torch.cuda.set_device(args.gpu if isinstance(args.gpu, int) else args.gpu[0])
        Model = models.get_model_by_name(args.model_name)
        DataLoader = dataloaders.get_dataloader_by_name(args.data_loader)
        model_hps = hp.combine_hparams_into_one(Model.default_hparams(), DataLoader.default_hparams())
        hp.load_config(model_hps, Model.get_config_filepath(args.output_dir, args.model_id))
        if args.model_hparams:
            model_hps.parse(args.model_hparams)
        dataset = DataLoader(model_hps, args.dataset)
        model = Model(model_hps, dataset, args.output_dir, args.model_id)
        model.restore_checkpoint_if_exists(args.resume)
    experiment = Experiment(Experiment.parse_hparams(args.exp_hparams), args.id, args.output_dir)
    print(experiment.compute(model))


if __name__ == '__main__':
    main()
