"""Command line driver with the reference's flags, prints and result file (reference: main.py).

    python -m coldrec_amd.main --dataset movielens --model MF --emb_size 64 --cold_object item

reads ./data/<dataset>/cold_<object>/{warm_train,overall_val,...}.csv + info_dict.pkl (+ content
.npy), trains on the MI355X and appends the text + JSON block to <result_dir>/<model>/<log>.
``python -m coldrec_amd.main --make_synthetic movielens`` first writes a synthetic dataset of
that published shape (no dataset ships with the reference).
"""
import argparse
import json
import os
import pickle
from datetime import datetime

import numpy as np
import torch

from .config.model_param import _str2bool, model_specific_param
from .model import AVAILABLE_MODELS
from .util.databuilder import ColdStartDataBuilder
from .util.loader import DataLoader
from .util.utils import set_seed

SETTINGS = [('Overall', 'all'), ('Cold-Start', 'cold'), ('Warm-Start', 'warm')]
METRICS = ['hit', 'precision', 'recall', 'ndcg']


class Config:
    """args + device + the ColdStartDataBuilder, built once and shared by all runs (main.py:15-58)."""

    def __init__(self, args: argparse.Namespace, data_root: str = './data'):
        self.args = args
        self.device = torch.device("cuda:%d" % args.gpu_id if (torch.cuda.is_available() and args.use_gpu) else "cpu")
        base = os.path.join(data_root, args.dataset, f'cold_{args.cold_object}')
        files = ['warm_train', 'warm_val', f'cold_{args.cold_object}_val', 'overall_val', 'warm_test',
                 f'cold_{args.cold_object}_test', 'overall_test']
        parts = [DataLoader.load_pairs(os.path.join(base, f + '.csv')) for f in files]
        with open(os.path.join(base, 'info_dict.pkl'), 'rb') as f:
            info = pickle.load(f)
        print(f"Dataset: {args.dataset}, User num: {info['user_num']}, Item num: {info['item_num']}.")
        content = np.load(os.path.join(data_root, args.dataset, f'{args.dataset}_{args.cold_object}_content.npy'))
        print(f'{args.cold_object} content shape: {content.shape}')
        self.data = ColdStartDataBuilder(
            parts[0], parts[1], parts[2], parts[3], parts[4], parts[5], parts[6], info['user_num'],
            info['item_num'], info['warm_user'], info['warm_item'], info['cold_user'], info['cold_item'],
            content if args.cold_object == 'user' else None, content if args.cold_object == 'item' else None)


def model_factory(config):
    cls = AVAILABLE_MODELS.get(config.args.model)
    if cls is None:
        raise ValueError(f"Invalid model name: {config.args.model}. Available models: {list(AVAILABLE_MODELS.keys())}")
    return cls(config)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', default='citeulike')
    p.add_argument('--model', default='MF')
    p.add_argument('--epochs', type=int, default=500)
    p.add_argument('--layers', type=int, default=2)
    p.add_argument('--topN', default='10,20')
    p.add_argument('--bs', type=int, default=4096, help='training batch size')
    p.add_argument('--emb_size', type=int, default=64)
    p.add_argument('--lr', type=float, default=0.001)
    p.add_argument('--reg', type=float, default=0.0001)
    p.add_argument('--runs', type=int, default=1, help='model runs')
    p.add_argument('--seed', type=int, default=2024)
    p.add_argument('--use_gpu', type=_str2bool, nargs='?', const=True, default=True)
    p.add_argument('--save_emb', type=_str2bool, nargs='?', const=True, default=True)
    p.add_argument('--gpu_id', type=int, default=0)
    p.add_argument('--cold_object', default='item', type=str, choices=['user', 'item'])
    p.add_argument('--backbone', default='MF')
    p.add_argument('--early_stop', type=int, default=10)
    p.add_argument('--eval_every', type=int, default=1)
    p.add_argument('--result_dir', type=str, default='./result')
    p.add_argument('--result_log', type=str, default='history.txt')
    p.add_argument('--result_file', type=str, default='')
    p.add_argument('--result_overwrite', action='store_true')
    p.add_argument('--data_root', type=str, default='./data', help='(addition) where <dataset>/ lives')
    p.add_argument('--lazy_adam', choices=['auto', 'on', 'off'], default='auto',
                   help='(addition) BPR-MF only: replay dense Adam on the rows a batch touches (bit-identical to the '
                        'dense pass, which it replaces when the tables are much larger than a batch)')
    p.add_argument('--optimizer', choices=['adam', 'sgd'], default='adam',
                   help='(addition) MF / LightGCN: adam = torch.optim.Adam as the reference (model/MF.py:14); sgd = '
                        'torch.optim.SGD(lr) defaults, the "BPR loss + SGD update" mode: rows a batch does not touch do '
                        'not move, 8 instead of 24 bytes of optimiser traffic per element')
    p.add_argument('--shard_graph', action='store_true',
                   help='(addition) LightGCN under torch.distributed.run: row-shard the adjacency and every layer state '
                        'over the ranks (one all-gather per layer and direction) instead of replaying all SpMMs on every '
                        'rank; for graphs beyond one GPU')
    p.add_argument('--score_dtype', choices=['fp32', 'fp16'], default='fp32',
                   help='(addition) table precision of the fused full-catalogue ranking: fp32 = exact, the '
                        'reference\'s arithmetic; fp16 = half tables, fp32 accumulation (16x the MFMA rate)')
    p.add_argument('--make_synthetic', type=str, default='', help='(addition) write a synthetic dataset of '
                   'this published shape (movielens | citeulike | toy) under --data_root and exit')
    return p


def parse_args(argv=None) -> argparse.Namespace:
    parser = build_parser()
    args, _ = parser.parse_known_args(argv)
    parser = model_specific_param(args.model, parser, AVAILABLE_MODELS)
    return parser.parse_args(argv)


def _plain(ns):
    out = {}
    for k, v in sorted(vars(ns).items()):
        out[k] = v if isinstance(v, (int, float, str, bool)) or v is None else (list(v) if isinstance(v, (list, tuple)) else repr(v))
    return out


def summarise(results, top_ns, time_results, args):
    """Console summary + the text/JSON block appended to the result file (main.py:189-301)."""
    payload = {}
    for i, top_n in enumerate(top_ns):
        print("*" * 80)
        payload[str(top_n)] = {}
        for name, key in SETTINGS:
            print(f"Top-{top_n} {name} Test Performance:")
            st = {lab: (float(np.mean(results[key][m][i])), float(np.std(results[key][m][i])))
                  for lab, m in zip(['Hit', 'Precision', 'Recall', 'NDCG'], METRICS)}
            print(', '.join(f"{lab}@{top_n}: {mu:.4f}±{sd:.4f}" for lab, (mu, sd) in st.items()))
            payload[str(top_n)][key] = {lab: {'mean': mu, 'std': sd} for lab, (mu, sd) in st.items()}
    print("Efficiency Performance:")
    mean_t, std_t = float(np.mean(time_results)), float(np.std(time_results))
    print(f"Time: {mean_t:.4f}±{std_t:.4f} seconds per completed training epoch.")

    lines = ['=== ColdRec Run Result ===', f'timestamp: {datetime.now().isoformat(timespec="seconds")}',
             f'method: {args.model}', f'dataset: {args.dataset}', f'cold_object: {args.cold_object}',
             f'backbone: {args.backbone}', f'runs: {args.runs}', '', '--- Hyperparameters ---']
    lines += [f'{k}: {v}' for k, v in sorted(_plain(args).items())]
    lines += ['', '--- Test Metrics (mean ± std) ---']
    for top_n in top_ns:
        for name, key in SETTINGS:
            m = payload[str(top_n)][key]
            lines.append(f'Top-{top_n} {name}: ' + ', '.join(
                f"{lab}={m[lab]['mean']:.4f}±{m[lab]['std']:.4f}" for lab in ['Hit', 'Precision', 'Recall', 'NDCG']))
    eff = {'seconds_per_completed_epoch_mean': mean_t, 'seconds_per_completed_epoch_std': std_t}
    lines += ['', '--- Efficiency ---', f'seconds_per_completed_epoch_mean: {mean_t:.6f}',
              f'seconds_per_completed_epoch_std: {std_t:.6f}', '', '--- JSON (machine-readable) ---',
              json.dumps({'method': args.model, 'hyperparameters': _plain(args), 'metrics': payload,
                          'efficiency': eff}, indent=2, ensure_ascii=False)]
    path = os.path.abspath(args.result_file) if str(args.result_file).strip() else \
        os.path.join(os.path.abspath(args.result_dir), args.model, args.result_log)
    if os.path.dirname(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
    mode = 'w' if args.result_overwrite else 'a'
    with open(path, mode, encoding='utf-8') as f:
        if mode == 'a' and os.path.isfile(path) and os.path.getsize(path) > 0:
            f.write('\n' + '=' * 80 + '\n')
        f.write('\n'.join(lines) + '\n')
    print(f"Results written ({'overwrite' if args.result_overwrite else 'append'}) to: {path}")
    return payload


def main(argv=None):
    args = parse_args(argv)
    print(args)
    if args.make_synthetic:
        from .data.synth import make_dataset, write_dataset
        out = write_dataset(make_dataset(args.make_synthetic, args.cold_object), args.data_root, args.dataset)
        print('synthetic dataset written to', out)
        return None
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        # (addition) launched by torch.distributed.run, one rank per GPU of the node: training is
        # data-parallel and evaluation item-row-sharded over RCCL (coldrec_amd/train.py, eval.py);
        # every rank computes the same metrics, rank 0 alone writes the result file and the tables
        import torch.distributed as dist
        args.gpu_id = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(args.gpu_id)
        dist.init_process_group('nccl', device_id=torch.device('cuda', args.gpu_id))
        if dist.get_rank() != 0:
            args.save_emb = False
    config = Config(args, args.data_root)
    top_ns = args.topN.split(',')
    results = {s: {m: [[] for _ in top_ns] for m in METRICS} for _, s in SETTINGS}
    time_results = []
    for round_num in range(args.runs):
        print(f"Start round {round_num} running!")
        set_seed(args.seed if args.runs == 1 else round_num, args.use_gpu)
        model = model_factory(config)
        print(f"Registered model: {args.model}.")
        model.run()
        for i in range(len(top_ns)):
            for key, res in (('all', model.overall_test_results), ('cold', model.cold_test_results),
                             ('warm', model.warm_test_results)):
                for q, m in enumerate(METRICS):
                    results[key][m][i].append(res[i][q])
        done = int(getattr(model, 'epochs_ran', 0) or 0) or (args.epochs if args.epochs > 0 else 1)
        time_results.append((model.train_end_time - model.train_start_time) / done)
    if world > 1:
        import torch.distributed as dist
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return None
    return summarise(results, top_ns, time_results, args)


if __name__ == '__main__':
    main()
