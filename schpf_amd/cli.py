"""`scHPF prep | prep-like | train | train-pool | score | project`: command-line entry points over
the loaders, run_trials and the model methods -- argument plumbing only, no logic of its own.

Same sub-commands, options, defaults and output file names as the reference's script
(/root/reference/bin/scHPF:40-292 options; :317-366 prep / prep-like, :370-470 train, :485-534 score,
:536-559 project),
so a pipeline that calls `scHPF train -i X.mtx -o out -k 7 -t 5` keeps working and finds
`out/scHPF_K7_b0_5trials.joblib` (the reference appends `_b{batchsize}` whenever ncells > batchsize,
hence also for batchsize 0).
"""
import argparse
import json
import os
import sys
import time
from functools import partial

import joblib
import numpy as np


def _add_train_options(train, nargs_k=None):
    train.add_argument("-i", "--input", required=True,
                       help="Training data: the .mtx written by `prep`, or a tab-separated "
                            "CELL_ID GENE_ID UMI_COUNT file (0-indexed, no duplicates).")
    train.add_argument("-o", "--outdir", help="Output directory (created if missing).")
    train.add_argument("-p", "--prefix", default="", help="Prefix for output files.")
    if nargs_k:
        train.add_argument("-k", "--nfactors", nargs="+", type=int, required=True, help="Numbers of factors.")
    else:
        train.add_argument("-k", "--nfactors", type=int, required=True, help="Number of factors.")
    train.add_argument("-t", "--ntrials", type=int, default=1, help="Random restarts; the best loss wins. [1]")
    train.add_argument("-v", "--validation-cells", default=None,
                       help="Held-out cells (same format as --input) used for convergence and model choice.")
    train.add_argument("-M", "--max-iter", type=int, default=1000, help="Maximum iterations. [1000]")
    train.add_argument("-m", "--min-iter", type=int, default=30, help="Minimum iterations. [30]")
    train.add_argument("-e", "--epsilon", type=float, default=0.001,
                       help="Minimum percent decrease of the loss between checks to continue. [0.001]")
    train.add_argument("-f", "--check-freq", type=int, default=10, help="Iterations between checks. [10]")
    train.add_argument("--better-than-n-ago", default=5, type=int,
                       help="Stop when the loss is worse than this many checks ago and rising. [5]")
    train.add_argument("-a", type=float, default=0.3, help="Hyperparameter a (-2: 1/sqrt(nfactors)). [0.3]")
    train.add_argument("-c", type=float, default=0.3, help="Hyperparameter c (-2: 1/sqrt(nfactors)). [0.3]")
    train.add_argument("--float32", action="store_true", help="32-bit variational parameters.")
    train.add_argument("-bs", "--batchsize", default=0, type=int, help="Cells per training round (0: all).")
    train.add_argument("-sl", "--smooth-loss", default=1, type=int, help="Average the loss over this many checks.")
    train.add_argument("-bts", "--beta-theta-simultaneous", action="store_true",
                       help="Update beta and theta from the previous round's values.")
    train.add_argument("-sa", "--save-all", action="store_true", help="Save every trial.")
    train.add_argument("-rp", "--reproject", action="store_true",
                       help="Reproject the data onto the fixed gene parameters before model selection.")
    train.add_argument("--quiet", dest="verbose", action="store_false", default=True,
                       help="Do not print intermediate losses.")
    train.add_argument("--devices", type=int, nargs="+", default=None,
                       help="HIP device ordinals to spread the restarts over (this build's addition).")


def _parser():
    parser = argparse.ArgumentParser(prog="scHPF", description="scHPF on MI355X")
    sub = parser.add_subparsers(dest="cmd")

    text_or_loom = ("a whitespace-delimited genes x cells UMI count matrix with two leading columns of gene "
                    "attributes (ENSEMBL id, gene name), or a loom file with the row attribute `Accession` or `Gene`")
    prep = sub.add_parser("prep", help="Filter genes and write the training matrix.")
    prep.add_argument("-i", "--input", required=True, help="Input data: " + text_or_loom + ".")
    prep.add_argument("-o", "--outdir", help="Output directory (created if missing; default: the input's).")
    prep.add_argument("-p", "--prefix", default="", help="Prefix for output files.")
    prep.add_argument("-m", "--min-cells", type=float, default=0.01,
                      help="Keep genes observed in at least this many cells; a value in (0, 1) is a proportion "
                           "of the cells. [0.01]")
    prep.add_argument("-w", "--whitelist", default="",
                      help="Two-column file (ENSEMBL id, gene name): keep only these genes.")
    prep.add_argument("-b", "--blacklist", default="",
                      help="Two-column file (ENSEMBL id, gene name): drop these genes, also when whitelisted.")
    prep.add_argument("-nvc", "--n-validation-cells", type=int, default=0,
                      help="Hold out this many randomly selected cells for validation. [0]")
    prep.add_argument("-vgid", "--validation-group-ids", default=None,
                      help="Single-column file of cell group ids (np.loadtxt): validation cells are spread "
                           "about evenly over the groups.")
    prep.add_argument("--validation-max-group-frac", type=float, default=0.5,
                      help="With -vgid: the largest share of a group that may be held out. [0.5]")
    prep.add_argument("--filter-by-gene-name", default=False, action="store_true",
                      help="Match the white/blacklist by gene name instead of ENSEMBL id.")
    prep.add_argument("--no-split-on-dot", default=False, action="store_true",
                      help="Accepted for compatibility: the reference ignores it for `prep` (identifiers are "
                           "always compared without their '.version' suffix), and so does this command.")

    like = sub.add_parser("prep-like", help="Write a data set with the genes of another, in the same order.")
    like.add_argument("-i", "--input", required=True, help="Input data: " + text_or_loom + ".")
    like.add_argument("-r", "--reference", required=True,
                      help="Two-column file (ENSEMBL id, gene name), e.g. prep's genes.txt: the genes to select "
                           "from the input, in this order; all must be present.")
    like.add_argument("-o", "--outdir", required=True, help="Output directory (created if missing).")
    like.add_argument("-p", "--prefix", default="", help="Prefix for output files.")
    like.add_argument("--by-gene-name", default=False, action="store_true",
                      help="Match against the reference by gene name instead of ENSEMBL id.")
    like.add_argument("--no-split-on-dot", default=False, action="store_true",
                      help="Compare identifiers with their '.version' suffix.")

    train = sub.add_parser("train", help="Train a model (restarts run one after the other on one GPU, or "
                                         "spread over --devices).")
    _add_train_options(train)

    pool = sub.add_parser("train-pool", help="Train several numbers of factors / restarts concurrently.")
    _add_train_options(pool, nargs_k="+")
    pool.add_argument("--njobs", type=int, default=0, help="Concurrent trials (0: as many as devices).")

    score = sub.add_parser("score", help="Write cell scores, gene scores and ranked gene lists as text.")
    score.add_argument("-m", "--model", required=True, help="A .joblib model written by `train`.")
    score.add_argument("-o", "--outdir", default=None,
                       help="Output directory (default: a directory named after the model file).")
    score.add_argument("-p", "--prefix", default="", help="Prefix for output files.")
    score.add_argument("-g", "--genefile", default=None,
                       help="Tab-delimited gene table without header (prep's genes.txt): also write ranked genes.")
    score.add_argument("--name-col", type=int, default=1, help="Zero-indexed column of --genefile with the names. [1]")

    proj = sub.add_parser("project", help="Project new cells onto a trained model.")
    proj.add_argument("-m", "--model", required=True, help="The model to project onto.")
    proj.add_argument("-i", "--input", required=True, help="Data to project (same formats as `train -i`).")
    proj.add_argument("-o", "--outdir", help="Output directory (default: the model's).")
    proj.add_argument("-p", "--prefix", default="", help="Prefix for output files.")
    proj.add_argument("--recalc-bp", action="store_true", help="Recompute the hyperparameter bp for the new data.")
    proj.add_argument("--max-iter", type=int, default=500, help="[500]")
    proj.add_argument("--min-iter", type=int, default=10, help="[10]")
    proj.add_argument("--epsilon", type=float, default=0.001, help="[0.001]")
    proj.add_argument("--check-freq", type=int, default=10, help="[10]")
    return parser


def _load_matrix(path):
    from .preprocessing import load_coo, load_mtx
    return load_mtx(path) if path.endswith(".mtx") else load_coo(path)


def _write_args(args, path):
    with open(path, "w") as fh:
        json.dump(args.__dict__, fh, indent=2)


def _write_matrix(path, X):
    from scipy.io import mmwrite
    mmwrite(path, X, field="integer")


def _prep(args, outprefix):
    """bin/scHPF:317-349: filtered.mtx, genes.txt, optionally the train / validation split."""
    from .preprocessing import load_and_filter, split_validation_cells
    filtered, genes = load_and_filter(args.input, min_cells=args.min_cells, whitelist=args.whitelist,
                                      blacklist=args.blacklist, filter_by_gene_name=args.filter_by_gene_name,
                                      no_split_on_dot=args.no_split_on_dot)
    print("Writing filtered data to file.....")
    _write_matrix("{}filtered.mtx".format(outprefix), filtered)
    genes.to_csv("{}genes.txt".format(outprefix), sep="\t", header=None, index=None)
    if args.n_validation_cells > 0:
        print("Selecting train/validation cells.....")
        Xtrn, Xvld, vld_ix = split_validation_cells(filtered, args.n_validation_cells, args.validation_group_ids,
                                                    max_group_frac=args.validation_max_group_frac)
        trn_ix = np.setdiff1d(np.arange(filtered.shape[0]), vld_ix)
        print("Writing train/validation splits.....")
        _write_matrix("{}train_cells.mtx".format(outprefix), Xtrn)
        np.savetxt("{}train_cell_ix.txt".format(outprefix), trn_ix, fmt="%d")
        _write_matrix("{}validation_cells.mtx".format(outprefix), Xvld)
        np.savetxt("{}validation_cell_ix.txt".format(outprefix), vld_ix, fmt="%d")
    print("Writing commandline arguments to file.....")
    _write_args(args, "{}prep_commandline_args.json".format(outprefix))


def _prep_like(args, outprefix):
    """bin/scHPF:353-366."""
    from .preprocessing import load_like
    print("Loading and reordering input like reference.... ")
    filtered, genes = load_like(args.input, reference=args.reference, by_gene_name=args.by_gene_name,
                                no_split_on_dot=args.no_split_on_dot)
    print("Writing prepared data to file.....")
    _write_matrix("{}filtered.mtx".format(outprefix), filtered)
    genes.to_csv("{}genes.txt".format(outprefix), sep="\t", header=None, index=None)
    print("Writing commandline arguments to file.....")
    _write_args(args, "{}prep-like_commandline_args.json".format(outprefix))


def _train(args, outprefix):
    from .trials import run_trials, run_trials_pool
    print("Loading data.....")
    train = _load_matrix(args.input)
    ncells, ngenes = train.shape
    print(".....found {} cells and {} genes in {}".format(ncells, ngenes, args.input))
    if args.batchsize and ncells > args.batchsize and not args.reproject:
        print("\nWARNING: running with minibatches but without reproject. We recommend adding the "
              "--reproject flag when running with batches to synchronize cell variational distributions. \n")
    vcells = None
    if args.validation_cells is not None:
        vcells = _load_matrix(args.validation_cells)
        print(".....found {} validation cells and {} genes in {}".format(vcells.shape[0], vcells.shape[1],
                                                                          args.validation_cells))
    print("Running trials.....")
    dtype = np.float32 if args.float32 else np.float64
    pooled = args.cmd != "train"
    if args.cmd == "train":
        run = run_trials
        if args.devices and len(args.devices) > 1:      # restarts spread over several GPUs
            run, pooled = partial(run_trials_pool, devices=args.devices, njobs=len(args.devices)), True
        elif args.devices:
            run = partial(run_trials, device=args.devices[0])
    else:
        if args.njobs < 0:
            raise ValueError("njobs must be an int >= 0, received {}".format(args.njobs))
        run = partial(run_trials_pool, njobs=args.njobs, devices=args.devices)
    result = run(train, vcells=vcells, nfactors=args.nfactors, ntrials=args.ntrials, min_iter=args.min_iter,
                 max_iter=args.max_iter, check_freq=args.check_freq, epsilon=args.epsilon,
                 better_than_n_ago=args.better_than_n_ago, dtype=dtype, verbose=args.verbose,
                 model_kwargs=dict(a=args.a, c=args.c), return_all=args.save_all, reproject=args.reproject,
                 batchsize=args.batchsize, beta_theta_simultaneous=args.beta_theta_simultaneous,
                 loss_smoothing=args.smooth_loss)
    model, reject = result if args.save_all else (result, None)
    klist = [args.nfactors] if isinstance(args.nfactors, int) else args.nfactors
    if not pooled:                                      # run_trials: one model (and one list of rejects)
        model = [model]
        reject = [reject] if reject is not None else None
    for i, (K, m) in enumerate(zip(klist, model)):
        stem = "{}scHPF_K{}{}_{}trials".format(outprefix, K,
                                               "_b{}".format(args.batchsize) if ncells > args.batchsize else "",
                                               args.ntrials)
        if vcells is None:
            print("Saving best model ({} factors).....".format(K))
            joblib.dump(m, stem + ".joblib")
        else:
            print("Saving best model (training data, {} factors).....".format(K))
            joblib.dump(m, stem + ".train.joblib")
            print("Computing final validation projection ({} factors)....".format(K))
            joblib.dump(m.project(vcells, replace=False), stem + ".validation_proj.joblib")
        if args.save_all:
            for j, r in enumerate(reject[i]):
                joblib.dump(r, stem + "_reject{}.joblib".format(j + 1))
    cmdfile = "{}train_commandline_args.json".format(outprefix)
    if os.path.exists(cmdfile):
        cmdfile = "{}train_commandline_args.{}.json".format(outprefix, time.strftime("%Y%m%d-%H%M%S"))
    _write_args(args, cmdfile)


def _score(args, outprefix):
    from .util import max_pairwise_table, mean_cellscore_fraction_list
    print("Loading model.....")
    model = joblib.load(args.model)
    cell_score, gene_score = model.cell_score(), model.gene_score()
    print("Saving scores.....")
    np.savetxt(outprefix + "cell_score.txt", cell_score, delimiter="\t")
    np.savetxt(outprefix + "gene_score.txt", gene_score, delimiter="\t")
    with open(outprefix + "mean_cellscore_fraction.txt", "w") as fh:
        fh.write("nfactors\tmean_cellscore_fraction\n")
        for i, frac in enumerate(mean_cellscore_fraction_list(cell_score)):
            fh.write("{}\t{}\n".format(i + 1, frac))
    max_pairwise_table(gene_score, ntop_list=[50, 100, 150, 200, 250, 300, 350, 400, 450, 500]).to_csv(
        outprefix + "maximum_overlaps.txt", sep="\t", index=False)
    if args.genefile is not None:
        genes = np.loadtxt(args.genefile, delimiter="\t", dtype=str)
        if genes.ndim == 1:
            genes = genes[:, None]
        name_col = min(args.name_col, genes.shape[1] - 1)
        print(".....using {}'th column of genefile as gene label".format(name_col))
        ranks = np.argsort(gene_score, axis=0)[::-1]
        ranked = np.stack([genes[ranks[:, k], name_col] for k in range(gene_score.shape[1])]).T
        np.savetxt(outprefix + "ranked_genes.txt", ranked, fmt="%s", delimiter="\t")
    _write_args(args, "{}score_commandline_args.json".format(outprefix))


def _project(args, outprefix):
    print("Loading reference model.....")
    model = joblib.load(args.model)
    print("Loading data.....")
    data = _load_matrix(args.input)
    print("Projecting data.....")
    projection = model.project(data, replace=False, verbose=True, recalc_bp=args.recalc_bp,
                               min_iter=args.min_iter, max_iter=args.max_iter, check_freq=args.check_freq,
                               epsilon=args.epsilon)
    if args.recalc_bp:
        outprefix += "recalc_bp."
    joblib.dump(projection, "{}{}.proj.joblib".format(outprefix, args.model.rsplit(".", 1)[0].split("/")[-1]))
    _write_args(args, "{}project_commandline_args.json".format(outprefix))


def main(argv=None):
    parser = _parser()
    args = parser.parse_args(argv)
    if args.cmd is None:
        parser.print_help(sys.stderr)
        return 1
    if args.outdir is None:     # the reference's defaults (bin/scHPF:305-311)
        if args.cmd in ("prep", "prep-like", "train", "train-pool"):
            args.outdir = args.input.rsplit("/", 1)[0] if "/" in args.input else "."
        elif args.cmd == "project":
            args.outdir = args.model.rsplit("/", 1)[0] if "/" in args.model else "."
        else:
            args.outdir = args.model.split(".joblib")[0]
    if not os.path.exists(args.outdir):
        print("Creating output directory {} ".format(args.outdir))
        os.makedirs(args.outdir)
    prefix = args.prefix.rstrip(".") + "." if args.prefix else ""
    outprefix = args.outdir + "/" + prefix
    {"prep": _prep, "prep-like": _prep_like, "train": _train, "train-pool": _train, "score": _score,
     "project": _project}[args.cmd](args, outprefix)
    return 0


if __name__ == "__main__":
    sys.exit(main())
