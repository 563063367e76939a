"""Per-model command line flags (reference: config/model_param.py).  Only the models this package
builds are known here; flags of other ColdRec plugins stay in their own checkout."""
import argparse


def _str2bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ('1', 'true', 't', 'yes', 'y', 'on'):
        return True
    if s in ('0', 'false', 'f', 'no', 'n', 'off'):
        return False
    raise argparse.ArgumentTypeError(f'expected a boolean value, got {v!r}')


def model_specific_param(model_name, parser, available_models):
    if model_name not in available_models.keys():
        raise ValueError(f"Invalid model name: {model_name}. Available models: {list(available_models.keys())}")
    # MF and LightGCN take no flags beyond the common ones (--layers is common, main.py:94)
    if model_name == 'DropoutNet':      # config/model_param.py:242-255
        parser.add_argument('--n_dropout', type=float, default=0.5, help='Dropout rate of the network training')
        parser.add_argument('--dropoutnet_hidden1', type=int, default=200, help='DeepCF first hidden width')
        parser.add_argument('--dropoutnet_hidden2', type=int, default=100, help='DeepCF second hidden width')
    return parser
