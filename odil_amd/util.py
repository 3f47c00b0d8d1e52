"""Driver loops and run-time bookkeeping (reference src/odil/util.py).

On the hot path: `optimize_grad` (util.py:190-240), `optimize_newton` (util.py:152-187),
`optimize` (util.py:243-246).  Kept as thin Python host code like the reference; the
difference is that nothing is pulled to the host per epoch: `pinfo` converts its device
scalars lazily, only when a callback actually reads them (the reference's
`np.array(loss)` in core.py:1238-1240 is a device->host sync every epoch).
`make_callback` / `setup_outdir` / `add_arguments` keep the reference's flag names,
`pinfo` keys and the throughput formula (util.py:408-419); history / plotting / checkpoint
file formats are outside the hot path.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .history import History
from .optimizer import EarlyStopError, Optimizer, make_optimizer  # noqa: F401

# Log sink of `printlog` (the reference's module-level names: examples and tests assign them, reference util.py:13-36)
g_log_file = sys.stderr
g_log_echo = False


def assert_equal(first, second, msg=""):
    """ValueError with the reference's wording when the two values differ (reference util.py:15-17)."""
    if first == second:
        return
    raise ValueError("Expected equal '{:}' and '{:}'{}".format(first, second, msg))


def set_log_file(f=None, echo=None):
    """Redirects `printlog` to the open file `f`; `echo` also copies every line to stderr."""
    global g_log_file, g_log_echo
    g_log_file = g_log_file if f is None else f
    g_log_echo = g_log_echo if echo is None else echo


def printlog(*msg):
    """One line, space-separated like print(), to the log file (and to stderr when echoing), flushed at once so
    that `tail -f train.log` follows a run."""
    line = " ".join(str(part) for part in msg) + "\n"
    sinks = [g_log_file]
    if g_log_echo and g_log_file is not sys.stderr:
        sinks.insert(0, sys.stderr)
    for sink in sinks:
        sink.write(line)
        sink.flush()


class LazyPinfo(dict):
    """pinfo = {terms, names, norms, loss} whose device scalars become NumPy on first read."""

    @staticmethod
    def _conv(v):
        if isinstance(v, torch.Tensor):
            return np.array(v.detach().cpu().numpy())
        if isinstance(v, list):
            return [LazyPinfo._conv(a) for a in v]
        return v

    def __getitem__(self, key):
        v = LazyPinfo._conv(dict.__getitem__(self, key))
        dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]


# --------------------------------------------------------------------------------------
# Arguments: same flag names / defaults as reference util.py:70-149
# --------------------------------------------------------------------------------------
_ARGS = [
    ("--epochs", int, None, "Maximum epochs, defaults to product of plot_every and frames"),
    ("--every_factor", float, 1, "Multiplier for all *_every options"),
    ("--plot_every", int, 5, "Epochs between plots"),
    ("--report_every", int, 10, "Epochs between reports to stdout"),
    ("--history_every", int, 1, "Epochs between entries of training history"),
    ("--checkpoint_every", int, 0, "Epochs between checkpoints"),
    ("--frames", int, 10, "Frames to plot. Zero disables first frame."),
    ("--outdir", str, ".", "Output directory"),
    ("--optimizer", str, "adamn", "Optimizer"),
    ("--seed", int, 1000, "Seed for numpy.random and the backend"),
    ("--plot_title", int, 0, "Enable title in plots"),
    ("--plotext", str, "pdf", "Extension of plots"),
    ("--history_full", int, 0, "Number of epochs to write history at every point"),
    ("--montage", int, 1, "Run montage after plotting"),
    ("--double", int, None, "Double precision. Defaults to runtime.dtype"),
    ("--echo", int, 0, "Echo log to stderr"),
    ("--epoch_start", int, 0, "Initial value of epoch"),
    ("--frame_start", int, 0, "Initial value of frame"),
    ("--checkpoint", str, None, "Continue from checkpoint in state_*.pickle"),
    ("--checkpoint_train", str, None, "Continue from history in state_*_train.pickle"),
    ("--callback_update_state", int, 0, "Update state after callback"),
    ("--bfgs_m", int, 50, "History size for L-BFGS"),
    ("--bfgs_maxls", int, 50, "Max evaluations in line search"),
    ("--bfgs_pgtol", float, None, "Convergence tolerance for L-BFGS-B"),
    ("--adam_epsilon", float, None, "Parameter epsilon in Adam"),
    ("--adam_beta_1", float, None, "Parameter beta_1 in Adam"),
    ("--adam_beta_2", float, None, "Parameter beta_2 in Adam"),
    ("--multigrid", int, 0, "Use multigrid decomposition"),
    ("--dump_data", int, 1, "Dump data_*.pickle with every plot"),
    ("--jac_nsmp0", int, 50, "(unused) Jacobi optimizer option"),
    ("--jac_nsmp1", int, 1, "(unused) Jacobi optimizer option"),
    ("--jac_factor", float, 1, "(unused) Jacobi optimizer option"),
    ("--jac_epsilon", float, 1e-8, "(unused) Jacobi optimizer option"),
]


def add_arguments(parser):
    for flag, typ, default, hlp in _ARGS:
        parser.add_argument(flag, type=typ, default=default, help=hlp)
    parser.add_argument("--mg_interp", type=str, default="stack", choices=["conv", "stack"],
                        help="Multigrid interpolation method (both run the same HIP kernel)")
    parser.add_argument("--nn_initializer", type=str, default="legacy", choices=["legacy", "glorot", "lecun", "he"],
                        help="Initializer for weights of neural networks")


def setup_outdir(args, relpath_args=None):
    """Output directory, args.json, train.log, epoch bookkeeping, seeds (reference util.py:281-334)."""
    from . import runtime

    outdir = args.outdir
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "args.json"), "w") as f:
        env = {k: os.environ.get(k, "") for k in ["ODIL_BACKEND", "ODIL_JIT", "ODIL_MT", "ODIL_DTYPE", "ODIL_FUSE"]}
        d = dict(vars(args), **env, runtime_backend=runtime.backend_name, runtime_dtype=runtime.dtype_name,
                 runtime_jit=runtime.enable_jit, runtime_gpu=True)
        json.dump(d, f, sort_keys=True, indent=4, default=str)
    os.chdir(outdir)
    set_log_file(open("train.log", "w"), echo=args.echo)
    for k in relpath_args or []:
        if getattr(args, k):
            setattr(args, k, os.path.relpath(getattr(args, k), start=outdir))

    # the three cadences are given for every_factor = 1: stretched together, never below one epoch
    for cadence in ("plot_every", "history_every", "report_every"):
        value = getattr(args, cadence)
        if value is not None:
            setattr(args, cadence, max(1, round(value * args.every_factor)))
    if args.epochs is None:  # as many epochs as the requested number of frames needs
        args.epochs = args.frames * args.plot_every
    if args.seed is not None:
        np.random.seed(args.seed)
        runtime.get_mod().random.set_seed(args.seed)
    printlog(" ".join(sys.argv))


def get_memory_usage_kb():
    """Peak resident set size of this process (kB), as the reference reports it (util.py:41-55)."""
    import resource

    return int(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)


def get_gpu_memory_usage_kb():
    """(allocated, reserved) device memory of the caching allocator in kB."""
    if not torch.cuda.is_available():
        return 0, 0
    return torch.cuda.memory_allocated() // 1024, torch.cuda.memory_reserved() // 1024


def make_callback(problem, args=None, epoch_func=None, report_func=None, history_func=None, checkpoint_func=None,
                  plot_func=None):
    cbinfo = argparse.Namespace()
    cbinfo.walltime = 0
    cbinfo.epoch = 0
    cbinfo.time_callback = 0
    cbinfo.time_start = time.time()
    cbinfo.problem = problem
    cbinfo.args = args
    cbinfo.frame = 0
    cbinfo.history = History(csvpath="train.csv", warmup=1) if getattr(args, "history_every", 0) else None

    def callback(state, epoch, pinfo):
        args = cbinfo.args
        domain = problem.domain
        report = bool(args.report_every) and epoch % args.report_every == 0
        hist = cbinfo.history is not None and (epoch % args.history_every == 0 or epoch < (args.history_full or 0))
        plot = epoch % args.plot_every == 0 and bool(epoch or args.frames)
        checkpoint = bool(args.checkpoint_every) and epoch % args.checkpoint_every == 0
        if (report or hist or plot or checkpoint) and torch.cuda.is_available():
            # epochs are enqueued asynchronously (whole epochs as graph replays): wait for the device
            # BEFORE the callback clock starts, so that device time is attributed to the epochs
            torch.cuda.synchronize()
        t_in = time.time()
        cbinfo.task_report, cbinfo.task_history, cbinfo.task_plot, cbinfo.task_checkpoint = report, hist, plot, checkpoint
        cbinfo.pinfo = pinfo
        if isinstance(problem.tracers, dict):
            problem.tracers["epoch"] = epoch
        if epoch_func is not None:
            epoch_func(problem, state, epoch, cbinfo)
        now = time.time()
        cbinfo.time_callback += now - t_in
        t_in = now
        walltime = now - cbinfo.time_start - cbinfo.time_callback
        if report:
            printlog("\nepoch={:05d}".format(epoch))
            if pinfo and "norms" in pinfo:
                norms, names = pinfo["norms"], pinfo["names"]
                printlog("residual: " + ", ".join(
                    "{}:{:.5g}".format(name or str(i), float(np.array(norm)))
                    for i, (norm, name) in enumerate(zip(norms, names))))
            if report_func is not None:
                report_func(problem, state, epoch, cbinfo)
            if epoch > cbinfo.epoch:
                wte = (walltime - cbinfo.walltime) / (epoch - cbinfo.epoch)
                thr = np.prod(domain.cshape) / wte if wte > 0 else 0
            else:
                wte, thr = 0, 0
            gpu_used, gpu_pool = get_gpu_memory_usage_kb()
            printlog("memory: {:} MiB, gpu_used: {:} MiB, gpu_pool: {:} MiB".format(
                get_memory_usage_kb() // 1024, gpu_used // 1024, gpu_pool // 1024))
            printlog("walltime: {:.3f} s, walltime+callback: {:.3f} s, walltime/epoch: {:.3f} ms".format(
                walltime, walltime + cbinfo.time_callback, wte * 1000))
            printlog("throughput: {:.3f} Mcells/s".format(thr / 1e6))
            cbinfo.walltime, cbinfo.epoch = walltime, epoch
        if hist:
            h = cbinfo.history
            h.append("epoch", epoch)
            h.append("frame", cbinfo.frame)
            if pinfo and "norms" in pinfo:
                for i, (norm, name) in enumerate(zip(pinfo["norms"], pinfo["names"])):
                    h.append("norm_{:}".format(name or str(i)), np.array(norm))
            if pinfo and "loss" in pinfo:
                h.append("loss", np.array(pinfo["loss"]))
            if getattr(args, "linsolver_history", 0) and pinfo and "linsolver" in pinfo:
                for key, val in pinfo["linsolver"].items():
                    if isinstance(val, (int, float, str, np.floating)):
                        h.append("lin_" + key, val)
            h.append("walltime", float(np.round(walltime, 3)))
            gpu_used, gpu_pool = get_gpu_memory_usage_kb()
            h.append("memory", get_memory_usage_kb() // 1024)
            h.append("gpu_used", gpu_used // 1024)
            h.append("gpu_pool", gpu_pool // 1024)
            if history_func is not None:
                history_func(problem, state, epoch, h, cbinfo)
            h.write()
        if plot:
            if plot_func is not None:
                plot_func(problem, state, epoch, cbinfo.frame, cbinfo)
            cbinfo.frame += 1
        if checkpoint:
            if checkpoint_func is not None:
                checkpoint_func(problem, state, epoch, cbinfo)
            else:
                from .core import checkpoint_save

                path = "checkpoint_{:06d}.pickle".format(epoch)
                printlog(path)
                checkpoint_save(domain, state, path)
        cbinfo.time_callback += time.time() - t_in

    def next_active(epoch):
        """The first epoch after `epoch` at which this callback does anything (report, history, plot, checkpoint, a user
        hook): on the epochs before it a call is a no-op, so an optimizer may run them without calling."""
        args = cbinfo.args
        if epoch_func is not None:
            return epoch + 1
        e = epoch + 1
        while True:
            if ((bool(args.report_every) and e % args.report_every == 0)
                    or (cbinfo.history is not None and (e % args.history_every == 0 or e < (args.history_full or 0)))
                    or (e % args.plot_every == 0 and bool(e or args.frames))
                    or (bool(args.checkpoint_every) and e % args.checkpoint_every == 0)):
                return e
            e += 1

    callback.cbinfo = cbinfo
    callback.next_active = next_active
    return callback


# --------------------------------------------------------------------------------------
# Driver loops
# --------------------------------------------------------------------------------------
def _pinfo(loss, terms, names, norms):
    return LazyPinfo(terms=terms, names=names, norms=norms, loss=loss)


def _poisson_newton_step(problem, state, args, status):
    """Newton step of a problem already recognised as the Poisson stencil (fused.detect: f(u) = A u -
    rhs with the zero-Dirichlet Laplacian A): A delta = -f(u) straight from the fused residual and the
    geometric multigrid, without forming the seven coefficient arrays of the Jacobian and without
    re-recognising them (512^3: 75 ms of a 215 ms step).  None when the general route must be taken:
    other operators, multigrid-decomposed unknowns, damping, or a solver choice that is not multigrid
    (`direct` switches to multigrid beyond the dense factorisation's reach exactly as linsolver.solve does)."""
    from . import gmg, ops
    from .core import Field
    from .linsolver import DENSE_MAX_UNKNOWNS

    if state.initialized:
        problem.recognise(state)  # (without a callback no evaluation precedes the first step)
    ev = getattr(problem, "_fused", None)
    linsolver = getattr(args, "linsolver", "direct")
    if ev is None or ev.nlvl != 1 or len(state.fields) != 1 or not int(os.environ.get("ODIL_NEWTON_SHORTCUT", 1)):
        return None  # (ODIL_NEWTON_SHORTCUT=0: the general route eval_operator_grad -> linearize -> linsolver.solve)
    (field,) = state.fields.values()
    if not isinstance(field, Field) or getattr(args, "linsolver_damp", 0) or getattr(args, "linsolver_dampdiag", 0):
        return None
    n = field.array.numel()
    if not (linsolver == "multigrid" or (linsolver == "direct" and n > DENSE_MAX_UNKNOWNS)):
        return None
    u = field.array.contiguous()
    r, _ = ops.poisson_residual(u, ev.rhs, ev.h2, fu=ev.fu, loss=ev.loss)
    mixed = ev.dtype == torch.float64 and bool(int(os.environ.get("ODIL_GMG_MIXED", 0))) and all(n % 2 == 0 for n in ev.cshape)
    solver = ev.__dict__.get("_gmg_mixed" if mixed else "_gmg")
    if solver is None and mixed:  # (float64 residual operator, float32 cycles: gmg.solve_mixed)
        solver = ev.__dict__["_gmg_mixed"] = (gmg.PoissonGMG(ev.cshape, ev.h2, ev.dtype, ev.device, lite=True),
                                              gmg.PoissonGMG(ev.cshape, ev.h2, torch.float32, ev.device))
    elif solver is None:
        solver = ev.__dict__["_gmg"] = gmg.PoissonGMG(ev.cshape, ev.h2, ev.dtype, ev.device)
    tol = 1e-12 if linsolver == "direct" else getattr(args, "linsolver_tol", 1e-10)
    if mixed:
        b = ops.scale(r, -1.0, out=r)  # the evaluator's residual buffer is scratch: negate it in place
        delta = gmg.solve_mixed(solver[0], solver[1], b, tol=tol, maxiter=getattr(args, "linsolver_maxiter", None) or 60, status=status)
        return delta.reshape(-1)
    # A d = r is solved and u - d formed (the solver is linear: d = -delta) -- the residual need not be negated first
    # (||r||: the evaluation above reduced mean(r^2) into ev.loss already)
    d = solver.solve(r, tol=tol, maxiter=getattr(args, "linsolver_maxiter", None) or 60, status=status, copy=False,
                     b_meansq=ev.loss)
    ops.axpy(u, d, -1.0)  # (d: a work buffer of the solver, consumed here)
    if u.data_ptr() != field.array.data_ptr():
        field.array.copy_(u)
    return True


def optimize_newton(args, problem, state, callback=None, **kwargs):
    """x <- x + delta with (M^T M) delta = -M^T r per epoch (reference util.py:152-187)."""
    from .linsolver import solve

    domain = problem.domain

    def eval_pinfo(state):
        loss, _, terms, names, norms = problem.eval_loss_grad_device(state)
        return _pinfo(loss, terms, names, norms)

    opt = Optimizer(name="newton", displayname="Newton")
    printlog("Running {} optimizer".format(opt.displayname))
    if callback:  # (the evaluations around the steps exist for the report only, reference util.py:156-158, 180)
        callback(state, args.epoch_start, eval_pinfo(state))
    for epoch in range(args.epoch_start, args.epochs):
        opt.evals += 1
        linstatus = dict()
        delta = _poisson_newton_step(problem, state, args, linstatus)
        applied = delta is True  # (the recognised-Poisson step updates the state itself)
        negative = False  # (True: `delta` is d of M d = r and the update is x - d)
        if delta is None:
            vector, matrix = problem.linearize_device(state)
            # M d = r is solved and x - d formed: every route is a linear solve, whose result for -r is exactly -d (IEEE
            # arithmetic is symmetric in sign) -- the pass that negated r first is not needed; the result is consumed
            # before the next solve (no copy out of the solver's work buffers)
            delta = solve(matrix, vector.contiguous(), args, linstatus, getattr(args, "linsolver", "direct"), consume=True)
            negative = True
        if getattr(args, "linsolver_verbose", 0):
            printlog(linstatus)
        from . import ops
        from .core import Field

        fields = list(state.fields.values())
        if applied:
            pass
        elif len(fields) == 1 and type(fields[0]) is Field and torch.is_tensor(fields[0].array) \
                and fields[0].array.is_contiguous() and fields[0].array.numel() == delta.numel():

            ops.axpy(fields[0].array, delta.to(fields[0].array.dtype), -1.0 if negative else 1.0)  # x += delta in place (util.py:176-178)
        else:
            packed = domain.pack_state(state)
            domain.unpack_state(packed - delta if negative else packed + delta, state)
        if callback:  # one extra evaluation per step, for the report only (reference util.py:180)
            report = eval_pinfo(state)
            report["linsolver"] = linstatus
            callback(state, epoch + 1, report)
    return domain.arrays_from_state(state), argparse.Namespace(epochs=args.epochs, evals=args.epochs)


def make_loss_grad(problem, state):
    """The `loss_grad(arrays) -> (loss, grads, pinfo)` callable the optimizers drive (reference util.py:197-206),
    with this package's hooks attached: `fused_adam` (update inside the gradient launches), `graph_safe` /
    `graph_begin` / `refresh` / `graph_end` (epochs replayed as a hipGraph)."""
    domain = problem.domain

    def loss_grad(arrays):
        domain.arrays_to_state(arrays, state)
        loss, grads, terms, names, norms = problem.eval_loss_grad_device(state)
        return loss, grads, _pinfo(loss, terms, names, norms)

    def fused_adam(arrays, m, v, alpha, omb1, omb2, eps):
        """loss + gradient with the Adam update of the leading array(s) applied inside the fused
        gradient launch; None when the problem has no fused evaluator that can do it."""
        ev = getattr(problem, "_fused", None) or getattr(problem, "_traced", None)
        if ev is None or not hasattr(ev, "eval_loss_grad_adam"):
            return None
        domain.arrays_to_state(arrays, state)
        res = ev.eval_loss_grad_adam(state, m, v, alpha, omb1, omb2, eps)
        if res is None:
            return None
        loss, grads, terms, names, norms, done = res
        return loss, grads, _pinfo(loss, terms, names, norms), done

    def small_epochs(arrays, m, v, table, omb1, omb2, eps):
        """A runner of whole Adam epochs in ONE launch each call (small 1-D / 2-D Poisson problems,
        fused.PoissonEvaluator.small_epochs), or None when the problem is not of that kind.  `table`: device tensor of
        the run's step sizes; runner(first, count) runs the epochs that use table[first : first + count] and returns the
        report of the last of them (device scalars, read lazily)."""
        ev = getattr(problem, "_fused", None)
        if ev is None or not hasattr(ev, "small_plan"):
            return None
        heads = ev.small_plan(arrays, m, v)
        if heads is None:
            return None
        domain.arrays_to_state(arrays, state)
        losses, norms = torch.empty_like(table), torch.empty_like(table)

        def runner(first, count):
            ev.small_epochs(heads, table[first:first + count], losses[first:first + count], norms[first:first + count],
                            omb1, omb2, eps)
            k = first + count - 1
            return _pinfo(losses[k], [losses[k]], ev.names, [norms[k]])

        return runner

    loss_grad.fused_adam = fused_adam
    loss_grad.small_epochs = small_epochs
    # hipGraph replay of whole Adam epochs (optimizer._EpochGraph): possible when the evaluation is
    # made of this package's kernels only -- the generic path has its own graph (Problem(jit=True))
    # (outputs in parameter space: replayable when they run as the generated kernel of param_expr.py; the torch replay
    # of param_tape.py bakes the host scalars of the current epoch in)
    loss_grad.graph_safe = lambda: getattr(problem, "_fused", None) is not None or (
        getattr(problem, "_traced", None) is not None and getattr(problem._traced, "graph_ok", True) and (
            not getattr(problem._traced, "offgrid", None) or getattr(problem._traced, "par_outputs", None) is not None))

    def graph_hook(name):
        def call(*a):
            traced = getattr(problem, "_traced", None)
            if traced is not None:
                getattr(traced, name)(*a)

        return call

    # replayed epochs: host scalars of a traced operator travel as rows of a device table (stencil_jit.py)
    loss_grad.graph_begin, loss_grad.refresh, loss_grad.graph_end = (
        graph_hook("graph_begin"), graph_hook("graph_upload"), graph_hook("graph_end"))
    return loss_grad


def optimize_grad(args, optname, problem, state, callback=None, **kwargs):
    """Gradient-based optimisation (reference util.py:190-240)."""
    domain = problem.domain
    mod = domain.mod
    loss_grad = make_loss_grad(problem, state)

    def callback_wrap(arrays, epoch, pinfo):
        domain.arrays_to_state(arrays, state)
        callback(state, epoch, pinfo)
        if getattr(args, "callback_update_state", 0):
            new = domain.arrays_from_state(state)
            for i in range(len(new)):
                arrays[i] = new[i]

    # (a callback that tells when it next DOES something lets the optimizer run the epochs in between in one launch)
    if not getattr(args, "callback_update_state", 0):
        callback_wrap.next_active = getattr(callback, "next_active", None)

    for src, dst in [("bfgs_m", "m"), ("bfgs_pgtol", "pgtol"), ("bfgs_maxls", "maxls"), ("adam_epsilon", "epsilon"),
                     ("adam_beta_1", "beta_1"), ("adam_beta_2", "beta_2")]:
        if getattr(args, src, None) is not None:
            kwargs[dst] = getattr(args, src)
    opt = make_optimizer(optname, dtype=domain.dtype, mod=mod, **kwargs)
    printlog("Running {} optimizer".format(opt.displayname))
    arrays = domain.arrays_from_state(state)
    _, _, pinfo = loss_grad(arrays)
    if callback:
        callback(state, args.epoch_start, pinfo)
    arrays, optinfo = opt.run(
        arrays, loss_grad=loss_grad, epochs=args.epochs - args.epoch_start,
        callback=callback_wrap if callback else None, epoch_start=args.epoch_start, lr=args.lr, **kwargs,
    )
    domain.arrays_to_state(arrays, state)
    return arrays, optinfo


def optimize(args, optname, problem, state, callback, **kwargs):
    if optname == "newton":
        return optimize_newton(args, problem, state, callback, **kwargs)
    return optimize_grad(args, optname, problem, state, callback, **kwargs)
