"""G+D training orchestration for MI355X (interface of the reference's gans/trainer.py:44-567:
`Trainer(cfg)`, `.step(i)`, `.sample()`, `.save_checkpoint()`, same checkpoint keys).

Sequence of one iteration = the reference's Trainer.step (:247-482): G step -> D step -> lazy R1
every `lazy.gp` iterations -> G_ema update -> ADA p update every `lazy.ada` iterations -> scalars.
What is different is the plumbing, chosen for MI355X:
  * one process per GPU; gradients live in one flat fp32 buffer per model and are exchanged with
    ONE all-reduce each (parallel.FlatGradSync) instead of DDP buckets; rank-0's mutable G
    buffers travel as one packed broadcast; all logged scalars as one packed all-reduce;
  * no per-iteration host synchronisation unless `sync_scalars` asks for Python floats;
  * the angle grid is passed once ([1,2,H,W]) and broadcast inside the kernels instead of being
    repeat_interleave'd to the batch (trainer.py:91-93);
  * real batches: synthetic generator in HBM when no dataset is present (benchmarks).
"""
import copy
import gc
import os
from collections import defaultdict

import numpy as np
import torch
import torch.optim as optim

import dgv2_native
from gans import parallel
from gans.augment.adaptive_augment import AdaptiveAugment
from gans.context_manager import gradient_accumulation
from gans.coords import CoordBridge, synthetic_angle_grid
from gans.datasets.synthetic import SyntheticRangeImages
from gans.models.builder import build_discriminator, build_generator
from gans.models.loss import GANLoss
from gans.models.ops.common import filter2d
from gans.models.ops.native import input_grads_only, x3_auto
from gans.utils import set_requires_grad, tanh_to_sigmoid


@torch.no_grad()
def ema_inplace(ema_model, new_model, decay):
    """reference: trainer.py:30-41 -- parameters lerp, buffers copied; foreach-batched."""
    ep = [p for _, p in ema_model.named_parameters()]
    np_ = [p for _, p in new_model.named_parameters()]
    if ep and ep[0].is_cuda and all(p.dtype == torch.float32 and p.is_contiguous() for p in ep + np_):
        from gans.models.ops import native
        native.lerp_list(ep, np_, 1 - decay)   # p_ema <- decay * p_ema + (1 - decay) * p, 72 tensors per launch
    else:
        torch._foreach_mul_(ep, decay)
        torch._foreach_add_(ep, np_, alpha=1 - decay)
    eb = [b for _, b in ema_model.named_buffers()]
    nb = [b for _, b in new_model.named_buffers()]
    torch._foreach_copy_(eb, nb)


class Trainer:
    def __init__(self, cfg, sync_scalars=True):
        self.cfg = cfg
        self.rank = int(cfg.training.get("rank", 0))
        self.num_gpus = int(cfg.training.get("num_gpus", 1))
        local = int(os.environ.get("LOCAL_RANK", self.rank))
        self.device = torch.device("cuda", local)
        torch.cuda.set_device(self.device)
        self.sync_scalars = sync_scalars
        self.resolution = cfg.model.generator.synthesis_kwargs.resolution
        self.B = int(cfg.training.get("batch_size_per_gpu", cfg.training.batch_size // self.num_gpus))
        self.batch_size = int(cfg.training.batch_size)
        # gradient accumulation (reference: trainer.py:255-257, context_manager.py:21-35): a rank's share of the global
        # batch is processed in `num_accumulation` chunks of batch_size_per_gpu; gradients are exchanged once, after
        # the last chunk.  (train_gan.py:183-185 always derives batch_size_per_gpu = batch_size // num_gpus, i.e. one
        # chunk; a config that sets both keys gets the loop.)
        if self.batch_size % (self.B * self.num_gpus) != 0:
            raise ValueError(f"training.batch_size ({self.batch_size}) must be a multiple of batch_size_per_gpu "
                             f"({self.B}) x num_gpus ({self.num_gpus})")
        self.num_accumulation = self.batch_size // (self.B * self.num_gpus)
        # the relativistic objectives compare every logit with the mean logit of the OTHER class: the generator step
        # then also needs D(A(real)) (reference: trainer.py:262,279-287)
        self.use_real_in_g = cfg.training.gan_objective in ("ragan", "rahinge", "ralsgan")

        # models (rank 0's initial weights are broadcast, as DDP's constructor does)
        self.G = build_generator(cfg.model.generator).to(self.device)
        self.D = build_discriminator(cfg.model.discriminator).to(self.device)
        parallel.broadcast_module(self.G)
        parallel.broadcast_module(self.D)
        self.G_ema = copy.deepcopy(self.G).eval()
        self.A = AdaptiveAugment(p_init=cfg.training.augment.p_init, p_target=cfg.training.augment.p_target,
                                 kimg=cfg.training.augment.kimg, **cfg.training.augment.policy).to(self.device)
        angle_file = f"data/coords/{cfg.dataset.name}.npy"
        self.coord = CoordBridge(
            num_ring=self.resolution[0], num_points=self.resolution[1], min_depth=cfg.dataset.min_depth,
            max_depth=cfg.dataset.max_depth, angle_file=angle_file if os.path.exists(angle_file) else None,
            angle_array=None if os.path.exists(angle_file) else synthetic_angle_grid(self.resolution[0]),
        ).eval().to(self.device)
        for m in (self.G, self.G_ema, self.D, self.A, self.coord):
            m.requires_grad_(False)
        payload = {"fp32": None, "bf16": torch.bfloat16}[str(cfg.training.get("grad_payload", "fp32"))]
        # G's flat gradient buffer also carries rank 0's mutable buffers (ema_var, w_avg): the broadcast that DDP issues
        # before the D step's G forward rides in the gradient all-reduce (parallel.FlatGradSync.carry_buffers)
        self.g_sync = parallel.FlatGradSync(self.G, payload, carry_buffers=True)
        self._g_bufs_synced = False   # True while every rank's G buffers are known to equal rank 0's
        # D's backward in two pieces (training.overlap_d_reduce; default: whenever there is someone to exchange with):
        # the head's gradients -- the 65536 -> 512 Linear, 134 of 154 MB -- are complete after the first few kernels of
        # the backward pass and travel while the trunk's backward runs (what DDP's reverse-order buckets do for the
        # reference, trainer.py:76-79)
        ov = cfg.training.get("overlap_d_reduce", None)
        self.split_d = parallel.is_dist() if ov is None else bool(ov)
        self.split_d = self.split_d and hasattr(self.D, "head_parameters")
        self.d_sync = parallel.FlatGradSync(self.D, payload, first=self.D.head_parameters() if self.split_d else None)
        self.ddp_models = (self.g_sync, self.d_sync)
        self.auxin = {"angle": self.coord.angle}  # [1,2,H,W]; kernels broadcast it over the batch

        # data
        if cfg.dataset.name == "synthetic":
            self.train_dataset = None
            self.iter_train_loader = iter(SyntheticRangeImages(
                self.resolution, cfg.dataset.min_depth, cfg.dataset.max_depth, self.B, self.device,
                seed=cfg.random_seed + self.rank))
        elif cfg.dataset.name == "kitti_raw":
            # reference: trainer.py:98-119 -- KITTI Raw scans, rank-sharded infinite sampler; the projection runs on
            # the GPU in this process (gans/datasets/kitti.py), hence no worker processes
            from gans.datasets.kitti import KITTIRaw
            from gans.utils import InfiniteSampler
            if not os.path.isdir(cfg.dataset.root):
                raise FileNotFoundError(f"dataset.root={cfg.dataset.root} does not exist (dataset.name=kitti_raw); "
                                        "set dataset.name=synthetic for in-HBM synthetic scans")
            self.train_dataset = KITTIRaw(root=cfg.dataset.root, split="train", shape=self.resolution,
                                          min_depth=cfg.dataset.min_depth, max_depth=cfg.dataset.max_depth,
                                          device=self.device)
            self.train_dataset.datalist = [f for f in self.train_dataset.datalist if os.path.exists(f)] \
                if cfg.dataset.get("skip_missing", False) else self.train_dataset.datalist
            self.train_loader = torch.utils.data.DataLoader(
                self.train_dataset, batch_size=self.B, num_workers=0, shuffle=False, drop_last=True,
                sampler=InfiniteSampler(self.train_dataset, rank=self.rank, num_replicas=self.num_gpus,
                                        seed=cfg.random_seed + self.rank))
            self.iter_train_loader = iter(self.train_loader)
        else:
            raise NotImplementedError(f"dataset.name={cfg.dataset.name}: only kitti_raw and synthetic are built")

        # losses and lazy-regularisation corrected Adam (reference: trainer.py:121-171)
        self.adversarial_loss = GANLoss(cfg.training.gan_objective).to(self.device)
        self.lazy_gp = int(cfg.training.lazy.gp)
        self.lazy_ada = int(cfg.training.lazy.ada)
        self.gp_weight = 0.0
        ratio_G = ratio_D = 1.0
        if cfg.training.loss.get("gp", 0) > 0.0:
            self.gp_weight = float(cfg.training.loss.gp) * self.lazy_gp
            ratio_D = self.lazy_gp / (self.lazy_gp + 1.0)
        # path-length regulariser (reference: trainer.py:148-152,308-365; off in dusty_v2.yaml: loss.pl 0).  The reference
        # block cannot run (it passes `angles=` to a forward that takes `angle` and reads a "styles" output that does not
        # exist); what it sets out to do is built: see pl_fb.
        self.pl_weight = 0.0
        self.lazy_pl = int(cfg.training.lazy.get("pl", 4))
        self.pl_ema = torch.zeros((), device=self.device)
        if cfg.training.loss.get("pl", 0) > 0.0:
            if cfg.model.generator.arch != "dusty_v2":
                raise NotImplementedError("path-length regularisation is built for the dusty_v2 generator")
            self.pl_weight = float(cfg.training.loss.pl) * self.lazy_pl
            ratio_G = self.lazy_pl / (self.lazy_pl + 1.0)
        lg, ld = cfg.training.lr.generator, cfg.training.lr.discriminator
        # hipGraph replay of the step bodies (training.hip_graph: true); Adam must then keep its step
        # counters on the device
        self.use_graphs = bool(cfg.training.get("hip_graph", False))
        self._graphs, self._graph_warm = {}, {}
        self.g_sync.capture = self.d_sync.capture = self.use_graphs   # eager runs keep c10d's asynchronous reductions
        self.reuse_d_bank = os.environ.get("DGV2_NO_BANK_REUSE") is None
        self._d_bank_captured = self.d_bank_reused = False
        self.optim_G = optim.Adam(self.G.parameters(), lr=lg.alpha * ratio_G, capturable=self.use_graphs, fused=self.device.type == "cuda",
                                  betas=(float(lg.beta1) ** ratio_G, float(lg.beta2) ** ratio_G))
        self.optim_D = optim.Adam(self.D.parameters(), lr=ld.alpha * ratio_D, capturable=self.use_graphs, fused=self.device.type == "cuda",
                                  betas=(float(ld.beta1) ** ratio_D, float(ld.beta2) ** ratio_D))
        self.x_real = torch.empty(self.B, 1, *self.resolution, device=self.device)  # static input of the graphs
        # the device word the kernels report broken promises to (include/dgv2.h "Status words": the x_exact promise of D's
        # fp32 epilogue conv); it must exist before the first capture and is read wherever this class synchronises
        dgv2_native.status_word(self.device)
        # every random number of a step body from ONE launch (native.rng_fill, csrc/rng.hip: a Philox stream whose state
        # lives on the device, so captured bodies draw fresh numbers on every replay) instead of ~8 generator launches with
        # their scale / clamp companions per body and two philox-state fills per hipGraph replay.  training.native_rng:
        # false keeps torch's generator (same distributions, another stream).
        from gans.models import dusty_v2 as _v2
        from gans.models.ops import native as _native
        self.native_rng = (bool(cfg.training.get("native_rng", True)) and self.device.type == "cuda"
                           and isinstance(self.G, _v2.Generator))
        if self.native_rng:
            # (re)seeded from torch's seed of THIS moment and rewound, in place (captured bodies of an earlier Trainer keep
            # reading the same device words): a run is reproducible from torch.manual_seed / init_random_seed like the
            # reference's torch.randn path.  The STREAM is Philox4x32-10 with its own counter layout, not torch's
            # generator: same distributions (moment-tested), different numbers -- "parity unpinned" for the draws
            # themselves, which is why every parity test injects them.
            _native.rng_state(self.device, seed=torch.initial_seed())
            if cfg.dataset.name == "synthetic":   # the synthetic scans' two draws from the same stream, one launch
                self.iter_train_loader.native_rng = True
        self._body = None
        self._d_bank_fresh = False   # True between the G step's D forward and the next D optimizer step
        self._gp_scalar = torch.zeros((), device=self.device)   # the last R1 penalty (input of the captured tail exchange)

        # resume
        self.start_iteration = 0
        if cfg.training.get("resume") is not None:
            # the reference's checkpoints pickle cfg as an OmegaConf tree: read through the restricted loader
            from gans.pretrained import load_checkpoint
            sd = load_checkpoint(cfg.training.resume, map_location="cpu")
            self.start_iteration = sd["step"] // self.batch_size
            self.G.load_state_dict(sd["G"])
            self.D.load_state_dict(sd["D"])
            self.G_ema.load_state_dict(sd["G_ema"])
            self.A.load_state_dict(sd["A"])
            if self.pl_weight > 0.0 and "pl_ema" in sd:
                self.pl_ema.copy_(sd["pl_ema"].to(self.device).reshape(()))
            self.optim_G.load_state_dict(sd["optim_G"])
            self.optim_D.load_state_dict(sd["optim_D"])
            if self.native_rng and "rng_state" in sd:   # continue the Philox stream where the saved run stopped
                _native.rng_state(self.device).copy_(sd["rng_state"].to(torch.int64))
            self.g_sync.rebind()
            self.d_sync.rebind()

        self.z_fixed = self.sample_z(self.B)
        self.warmup_fade_kimg = cfg.training.warmup.fade_kimg * 1e3
        self.blur_sigma = 0
        self.dropout_ratio = 0
        self.iters_to_imgs = lambda i: int(i * self.batch_size)
        # device-side warm-up schedule: the step bodies read the blur taps and the dropout ratio from these two static
        # buffers (refreshed by set_warmup_params), so the fade-in regime replays as hipGraphs too
        self._wu_n = int(np.floor(float(cfg.training.warmup.blur_init_sigma) * 3)) if self.warmup_fade_kimg > 0 else 0
        self._wu_taps = torch.zeros(2 * self._wu_n + 1, device=self.device)
        self._wu_ratio = torch.zeros((), device=self.device)
        self._wu_host = None
        self._injected = None   # static buffers of injected random draws (parity tests), see set_draws

    # ------------------------------------------------------------------ helpers
    def sample_z(self, batch_size):
        return torch.randn(batch_size, self.cfg.model.generator.mapping_kwargs.in_ch, device=self.device)

    def fetch_reals(self, raw_batch, out=None):
        """reference: trainer.py:211-217, fused into one kernel (written into `out` when given: the static batch the
        captured bodies read, without a copy behind the conversion)."""
        mask = raw_batch["mask"].to(self.device)
        x = self.coord.fetch_reals(raw_batch["depth"].to(self.device), mask, float(self.cfg.dataset.raydrop_const), out=out)
        return {"image": x, "raydrop_mask": mask}

    def set_warmup_params(self, iteration):
        """reference: trainer.py:219-232; also refreshes the device copies the step bodies read."""
        num_imgs = self.iters_to_imgs(iteration)
        w = self.cfg.training.warmup
        fade = max(1 - num_imgs / self.warmup_fade_kimg, 0) if self.warmup_fade_kimg > 0 else 0
        self.blur_sigma = fade * w.blur_init_sigma
        self.dropout_ratio = fade * w.dropout_init_ratio
        if (self.blur_sigma, self.dropout_ratio) != self._wu_host:
            self._wu_host = (self.blur_sigma, self.dropout_ratio)
            n, nmax = int(np.floor(self.blur_sigma * 3)), self._wu_n
            taps = np.zeros(2 * nmax + 1, dtype=np.float32)
            if n > 0:   # trainer.py:238-239 + the normalisation of filter2d (common.py:29), zero beyond floor(3 sigma)
                k = np.exp2(-np.square(np.arange(-n, n + 1, dtype=np.float32) / np.float32(self.blur_sigma)))
                taps[nmax - n:nmax + n + 1] = k / k.sum()
            else:
                taps[nmax] = 1.0
            self._wu_taps.copy_(torch.from_numpy(taps))
            self._wu_ratio.fill_(float(self.dropout_ratio))

    def _warm(self):
        return self.blur_sigma > 0 or self.dropout_ratio > 0

    def warmup(self, x, keep=None):
        """reference: trainer.py:234-245.  Static shapes: the blur always has 2 floor(3 sigma_init) + 1 taps (zeros
        beyond the current floor(3 sigma): the ring / replicate extension they touch is weighted by zero) and the
        Bernoulli keep mask compares against the device-side ratio.  `keep` injects the mask (parity tests)."""
        if not self._warm():
            return x
        if self._wu_n > 0:
            x = filter2d(x, self._wu_taps, normalize=False)
        if keep is None:
            if float(self.cfg.training.warmup.dropout_init_ratio) == 0.0:
                return x                      # blur-only warm-up: no mask to draw, nothing to blend
            keep = (torch.rand_like(x) < (1 - self._wu_ratio)).to(x.dtype)
        return keep * x + (1 - keep) * float(self.cfg.dataset.raydrop_const)

    # ------------------------------------------------------------------ injected randomness (parity tests)
    def set_draws(self, draws):
        """Inject the random numbers of the NEXT iteration, keyed by call site as in tests/golden/make_golden.py
        (TRAINER_SITES): g.z, g.shifts, g.u, g.keep, g.ada.{G,C}, d.z, d.shifts, d.u, d.keep_real, d.keep_fake,
        d.ada_real.{G,C}, d.ada_fake.{G,C}, r1.keep, r1.ada.{G,C}.  The values are copied into static device buffers
        that the step bodies read, so injected runs replay as hipGraphs as well.  None restores on-device sampling."""
        if draws is None:
            self._injected = None
            return
        if self._injected is None:
            self._injected = {}
        for k, v in draws.items():
            v = v.to(self.device, torch.float32)
            if k not in self._injected:
                self._injected[k] = v.clone()
            else:
                self._injected[k].copy_(v)

    def _begin_body(self, body):
        """Draw every random number the body `body` (g / d / r1) will consume with one launch (self.native_rng); the
        helpers below hand them out by call site.  Injected draws (set_draws) take precedence."""
        self._body = None
        if self._injected is not None or not self.native_rng:
            return
        from gans.models.ops import native
        B, (H, W) = self.B, self.resolution
        syn = self.G.synthesis_network
        eps = float(torch.finfo(torch.float32).eps)
        specs, keys = [], []

        def add(key, shape, kind, a, b):
            keys.append(key)
            specs.append((shape, kind, a, b))

        if body in ("g", "d"):
            add(body + ".z", (B, int(self.cfg.model.generator.mapping_kwargs.in_ch)), native.RNG_NORMAL, 0.0, 1.0)
            if self.G.training and syn.aug_coords and not syn.aug_coords_blitting:
                add(body + ".shifts", (B,), native.RNG_UNIFORM, 0.0, 2 * np.pi)   # dusty_v2.py:267-274
            add(body + ".u", (B, 1, H, W), native.RNG_CLAMPED, eps, 1.0 - eps)     # gumbel.py:23-29 (clamp_probs)
        sites = {"g": ["g.ada"] + (["g.ada_real"] if self.use_real_in_g else []),
                 "d": ["d.ada_real", "d.ada_fake"], "r1": ["r1.ada"]}[body]
        for site in sites:
            add(site + ".u", (B, 16), native.RNG_UNIFORM, 0.0, 1.0)                # adaptive_augment.py:386-470
            add(site + ".n", (B, 8), native.RNG_NORMAL, 0.0, 1.0)
        if self._warm() and float(self.cfg.training.warmup.dropout_init_ratio) > 0.0:
            ksites = {"g": ["g.keep"] + (["g.keep_real"] if self.use_real_in_g else []),
                      "d": ["d.keep_real", "d.keep_fake"], "r1": ["r1.keep"]}[body]
            for site in ksites:
                add(site + ".uniform", (B, 1, H, W), native.RNG_UNIFORM, 0.0, 1.0)  # trainer.py:241-245
        self._body = dict(zip(keys, native.rng_fill(specs, self.device)))

    def _draw(self, key):
        if self._injected is not None:
            return self._injected.get(key)
        if self._body is not None and key + ".uniform" in self._body:   # a keep mask from this body's uniforms
            return (self._body[key + ".uniform"] < (1 - self._wu_ratio)).float()
        return None

    def _z(self, site):
        z = self._draw(site + ".z")
        if z is None and self._body is not None:
            z = self._body.get(site + ".z")
        return self.sample_z(self.B) if z is None else z

    def _g_noise(self, site):
        if self._injected is not None:
            return {"shifts": self._injected[site + ".shifts"], "gumbel_u": self._injected[site + ".u"]}
        if self._body is not None:
            noise = {"gumbel_u": self._body[site + ".u"]}
            if site + ".shifts" in self._body:
                noise["shifts"] = self._body[site + ".shifts"]
            return noise
        return None

    def _ada(self, site):
        if self._injected is not None:
            if site + ".G" not in self._injected:
                return None
            return {"G": self._injected[site + ".G"], "C": self._injected[site + ".C"]}
        if self._body is not None and site + ".u" in self._body:
            return {"u": self._body[site + ".u"], "n": self._body[site + ".n"]}
        return None

    # ------------------------------------------------------------------ sub-steps
    # Each sub-step is split into a forward/backward body (`*_fb`), the gradient all-reduce (eager,
    # RCCL) and the optimizer body, so that the bodies can be replayed as hipGraphs while the
    # collectives stay ordinary stream work between them.
    def g_fb(self, j, scalars, x_real=None, pack=True):
        set_requires_grad(self.G, True)
        self.g_sync.begin(direct=j == 0)
        self._begin_body("g")
        z = self._z("g")
        x_fake = self.G(z, noise=self._g_noise("g"), **self.auxin)["image"]
        y_fake = self.D(self.A(self.warmup(x_fake, self._draw("g.keep")), draws=self._ada("g.ada")))
        # D's weight bank now matches D's weights (until the next D optimizer step).  WHERE it lives decides who may
        # read it again: built inside a capture it sits in this body's graph pool and every replay rewrites it in place;
        # built eagerly it is an ordinary allocation that the next eager G step replaces (and frees).
        self._d_bank_fresh = True
        self._d_bank_captured = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        y_real = None
        if x_real is not None:   # relativistic objectives only (trainer.py:279-285: the augmented reals are detached)
            with torch.no_grad():
                x_real_aug = self.A(self.warmup(x_real, self._draw("g.keep_real")), draws=self._ada("g.ada_real"))
            y_real = self.D(x_real_aug)
        if self.adversarial_loss.can_fuse(y_fake):
            # softplus(-y_fake).mean() and the cotangent loss.gan * d loss / d y from one launch; y.backward(gy) replaces
            # (loss.gan * loss).backward() and its five scalar-graph launches
            stats, gy = self.adversarial_loss.fused_nsgan_step(y_fake, len(y_fake), float(self.cfg.training.loss.gan))
            loss_gan = stats[0]
            y_fake.backward(gy)
        else:
            loss_gan = self.adversarial_loss(y_real, y_fake, "G")
            (self.cfg.training.loss.gan * loss_gan).backward()
        self.g_sync.collect(accumulate=j > 0, scale=1.0 / self.num_accumulation, pack=pack)
        set_requires_grad(self.G, False)
        scalars["loss/G/adversarial"] = loss_gan.detach()

    # One process, one chunk: nothing happens between a body's backward and its optimizer step (no exchange, no second
    # chunk), so both are ONE body -- one graph launch instead of two, and Adam reads the gradients where the backward
    # left them (FlatGradSync.collect(pack=False): no pack copy, 40 us of multi-tensor launches for G's 140 tensors).
    def g_fb_opt(self, j, scalars):
        self.g_fb(j, scalars, pack=False)
        self._opt_step(self.optim_G)

    def g_fb_rel_opt(self, x_real, j, scalars):
        self.g_fb(j, scalars, x_real, pack=False)
        self._opt_step(self.optim_G)

    def d_fb_opt(self, x_real, j, scalars):
        self.d_fb(x_real, j, scalars, pack=False)
        self._opt_step(self.optim_D)

    def r1_fb_opt(self, x_real, j, scalars):
        self.r1_fb(x_real, j, scalars, pack=False)
        self._opt_step(self.optim_D)

    def g_fb_rel(self, x_real, j, scalars):
        """G step of a relativistic objective: also D(A(real)) (argument order of the other bodies that take reals)."""
        self.g_fb(j, scalars, x_real)

    def d_fb(self, x_real, j, scalars, cut=False, pack=True):
        set_requires_grad(self.D, True)
        self.d_sync.begin(direct=j == 0)
        self._begin_body("d")
        z = self._z("d")
        with torch.no_grad():
            x_fake = self.G(z, noise=self._g_noise("d"), **self.auxin)["image"]
            # both augmented halves are written straight into the stacked batch (no concatenation pass)
            x_both = torch.empty((2 * self.B,) + tuple(x_real.shape[1:]), device=self.device, dtype=torch.float32)
            self.A(self.warmup(x_real, self._draw("d.keep_real")), draws=self._ada("d.ada_real"), out=x_both[:self.B])
            self.A(self.warmup(x_fake, self._draw("d.keep_fake")), draws=self._ada("d.ada_fake"), out=x_both[self.B:])
        # D(real) and D(fake) in ONE pass over the discriminator (minibatch-stddev per half), instead of
        # the reference's two calls (trainer.py:391-392): same result, half the launches / weight reads
        # (the G step's D forward prepared the weight bank of these very weights: only G has moved since)
        # A captured D body bakes the bank's ADDRESSES in: it may reuse only a bank that a captured G body rewrites at
        # those addresses before every replay; an eager D body only a bank an eager G body just built.  A G body that
        # fell back to eager (failed capture, DGV2_GRAPHS) beside a captured D body -- or the reverse -- rebuilds.
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        reuse = {"reuse_bank": True} if (self.reuse_d_bank and self._d_bank_fresh and hasattr(self.D, "_bank_keep")
                                         and capturing == self._d_bank_captured) else {}
        self.d_bank_reused = bool(reuse)
        y = self.D(x_both, splits=2, **reuse, **({"cut": True} if cut else {}))
        y_real, y_fake = y[:self.B], y[self.B:]
        if self.adversarial_loss.can_fuse(y):
            # objective, its weighted gradient, both output means and ADA's sign statistic from one launch
            # (ADA's cumulate, adaptive_augment.py:368-370, rides in the same launch)
            stats, gy = self.adversarial_loss.fused_nsgan_step(y, self.B, float(self.cfg.training.loss.gan),
                                                               cum=(self.A.sign_cum, self.A.n_pred_cum))
            loss_gan, out_real, out_fake = stats[0], stats[1], stats[2]
            y.backward(gy)
        else:
            self.A.cumulate(y_real)
            loss_gan = self.adversarial_loss(y_real, y_fake, "D")
            out_real, out_fake = y_real.mean().detach(), y_fake.mean().detach()
            (self.cfg.training.loss.gan * loss_gan).backward()
        if cut:
            # the head's gradients are final; the trunk's backward is d_fb_tail
            self._d_cut = self.D.take_cut()
            self.d_sync.collect(accumulate=j > 0, scale=1.0 / self.num_accumulation, part="first")
        else:
            self.d_sync.collect(accumulate=j > 0, scale=1.0 / self.num_accumulation, pack=pack)
        scalars["loss/D/output/real"] = out_real
        scalars["loss/D/output/fake"] = out_fake
        scalars["loss/D/adversarial"] = loss_gan.detach()

    def d_fb_head(self, x_real, j, scalars):
        """D step up to and including the backward of the head (the two Linear layers behind Discriminator's cut)."""
        self.d_fb(x_real, j, scalars, cut=True)

    def d_fb_tail(self, j, scalars):
        """The rest of D's backward, from the gradient the head left at the cut.  Under hipGraphs this is a second
        capture: its kernels read the activations the first capture's replay wrote (each graph has its own pool, the
        saved tensors stay allocated in the first one's)."""
        feat, leaf = self._d_cut
        self._d_cut = None
        feat.backward(leaf.grad)
        self.d_sync.collect(accumulate=j > 0, scale=1.0 / self.num_accumulation, part="rest")

    def r1_fb(self, x_real, j, scalars, pack=True):
        """lazy R1 (reference: trainer.py:419-451): double backward through D and ADA."""
        set_requires_grad(self.D, True)
        self.d_sync.begin(direct=False)
        self._begin_body("r1")
        x = x_real.detach().requires_grad_(True)
        # behind a bf16 trunk the fp32 epilogue conv of this pass (forward, data gradient and the forward conv of the
        # double backward; no weight bank here) runs on the bf16 matrix cores too (conv_x3.hip, fp32-equivalent)
        conv = self.D.epilogue[1] if hasattr(self.D, "epilogue") else None
        live = {(conv.in_ch + 15) // 16 * 16: conv.in_ch} if conv is not None and hasattr(conv, "in_ch") else None
        with x3_auto(getattr(self.D, "num_fp16_layers", 0) == -1, live):
            return self._r1_body(x, j, scalars, pack)

    def _r1_body(self, x, j, scalars, pack=True):
        y_real = self.D(self.A(self.warmup(x, self._draw("r1.keep")), draws=self._ada("r1.ada")), double_backward=True)
        # (only the gradient w.r.t. the input is taken here: the Functions skip their weight / bias gradients, which
        # ctx.needs_input_grad alone would make them compute and discard)
        with input_grads_only():
            (grads,) = torch.autograd.grad(outputs=[y_real.sum()], inputs=[x], create_graph=True)
        r1 = (grads ** 2).sum(dim=[1, 2, 3]).mean()
        # The reference adds `0.0 * y_real.squeeze()[0]` (trainer.py:441) so that every parameter of D receives a
        # gradient under DDP: an exactly-zero cotangent pushed through D's forward graph from the top, i.e. a data and a
        # weight gradient OF ZEROS through the two Linear layers and the fp32 epilogue conv (the only real second-order
        # path into the forward graph, minibatch-stddev's backward, enters below them).  collect() zero-fills the slices
        # of parameters without a gradient, which is what the term produces for them: same flat gradient, same Adam
        # step.  (DGV2_R1_ZERO_TERM restores the literal form for A/B runs.)
        loss = (self.gp_weight / 2) * r1
        if os.environ.get("DGV2_R1_ZERO_TERM"):
            loss = loss + 0.0 * y_real.squeeze()[0]
        loss.backward()
        self.d_sync.collect(accumulate=j > 0, scale=1.0 / self.num_accumulation, pack=pack)
        scalars["loss/D/gradient_penalty"] = r1.detach()

    def pl_fb(self, j, scalars):
        """Path-length regulariser (what the reference's block sets out to do, trainer.py:308-365): half a batch of
        fakes, y ~ N(0, 1 / HW), J^T y = d(image . y) / dw for the styles w = mapping(z) [B/2, num_styles, D] (taken
        with create_graph through the generator's twice-differentiable pass, Generator.forward(second_order=True)),
        lengths |J^T y| per (sample, style), their running mean pl_ema (lerp 0.01), penalty mean((|J^T y| - pl_ema)^2)
        on the un-detached new mean exactly as the reference writes it (trainer.py:349-353), weighted by
        loss.pl * lazy.pl; its backward is the double backward through the generator."""
        set_requires_grad(self.G, True)
        self.g_sync.begin(direct=False)
        B_pl = max(self.B // 2, 1)
        z = self._z("pl")[:B_pl]
        w = self.G.forward_mapping(z)
        out = self.G(w, input_w=True, noise=self._g_noise_pl(B_pl), second_order=True, **self.auxin)
        image = out["image"]
        y = self._draw("pl.noise")
        y = torch.randn_like(image) if y is None else y[:B_pl]
        y = y / float(np.sqrt(np.prod(image.shape[2:])))
        (grads,) = torch.autograd.grad(outputs=[(image * y).sum()], inputs=[w], create_graph=True)
        lengths = grads.pow(2).sum(dim=-1).sqrt()
        # reference trainer.py:349-353: the NEW running mean enters the penalty un-detached (the gradient also flows
        # through its 0.01 * mean(lengths) term); only the stored buffer is detached
        ema_new = self.pl_ema.lerp(lengths.mean(), 0.01)
        self.pl_ema.copy_(ema_new.detach())
        penalty = (lengths - ema_new).pow(2).mean()
        loss = self.pl_weight * penalty + 0.0 * image[0, 0, 0, 0]
        loss.backward()
        self.g_sync.collect(accumulate=j > 0, scale=1.0 / self.num_accumulation)
        set_requires_grad(self.G, False)
        scalars["loss/G/path_length/baseline"] = self.pl_ema.detach().clone()
        scalars["loss/G/path_length"] = penalty.detach()

    def _g_noise_pl(self, n):
        if self._injected is None:
            return None
        return {"shifts": self._injected["pl.shifts"][:n], "gumbel_u": self._injected["pl.u"][:n]}

    def ema_decay(self, iteration):
        ema_imgs = int(self.cfg.training.ema_kimg * 1e3)
        if self.cfg.training.ema_rampup is not None:
            ema_imgs = min(ema_imgs, iteration * self.batch_size * self.cfg.training.ema_rampup)
        return 0.5 ** (self.batch_size / max(ema_imgs, 1e-8))

    # ------------------------------------------------------------------ hipGraph plumbing
    def _graphs_usable(self):
        return self.use_graphs

    def _run(self, name, fn, *args):
        """Run `fn(*args, scalars)` eagerly, or capture it once as a hipGraph and replay it.
        Returns the dict of scalar tensors the body produced (static buffers under replay)."""
        only = os.environ.get("DGV2_GRAPHS")  # debugging aid: comma list of bodies allowed to be graphs
        if not self._graphs_usable() or (only is not None and name not in only.split(",")):
            scalars = {}
            fn(*args, scalars)
            return scalars
        if self._warm() and not name.endswith("_opt"):
            name = name + "/warmup"   # the fade-in regime has its own captures (extra blur / dropout work)
        if self._injected is not None and not name.endswith("_opt"):
            name = name + "/inj"      # bodies that read injected draws are different graphs than the sampling ones
        if name not in self._graphs:
            if self._graph_warm.get(name, 0) < 2:  # allocator / autotune warm-up before capture
                self._graph_warm[name] = self._graph_warm.get(name, 0) + 1
                scalars = {}
                fn(*args, scalars)
                return scalars
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            scalars = {}
            # one private memory pool per graph: the bodies are replayed in a different order than they
            # were captured (lazy R1, shared optimizer graph), which a shared pool does not allow; HBM is
            # not the constraint here (288 GB)
            # thread_local: API calls of other threads (the RCCL watchdog polls events) must not invalidate the capture
            # no cyclic garbage collection while the stream is capturing: a collection (it can start on the autograd
            # thread) that frees device memory or a CUDAGraph of an earlier, unreachable Trainer issues calls the
            # capture forbids -- the destructor throws and the process aborts (torch.cuda.graph collects once on entry)
            gc_was_enabled = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn(*args, scalars)
            except Exception as e:  # keep training: this body runs eagerly from now on
                import warnings
                torch.cuda.synchronize()
                warnings.warn(f"hipGraph capture of '{name}' failed ({type(e).__name__}: {e}); running it eagerly")
                self._graphs[name] = None
                if gc_was_enabled:
                    gc.enable()
                scalars = {}
                fn(*args, scalars)
                return scalars
            finally:
                if gc_was_enabled:
                    gc.enable()
            self._graphs[name] = (g, scalars)
        if self._graphs[name] is None:
            scalars = {}
            fn(*args, scalars)
            return scalars
        g, scalars = self._graphs[name]
        g.replay()
        if os.environ.get("DGV2_GRAPH_SYNC"):  # debugging aid: serialise host and device after each replay
            torch.cuda.synchronize()
        return scalars

    def _link_graphs(self, head, tail):
        """The tail of a split body needs the autograd graph its head built: if either capture failed, both run
        eagerly from now on."""
        names = [n for n in self._graphs if n.split("/")[0] in (head.split("/")[0], tail.split("/")[0])]
        if any(self._graphs[n] is None for n in names):
            for n in names:
                self._graphs[n] = None

    def _opt_step(self, opt):
        if self.device.type == "cuda" and os.environ.get("DGV2_TORCH_ADAM") is None:
            from gans.models.ops import native
            native.fused_adam_step(opt)
        else:
            opt.step()

    # ------------------------------------------------------------------ one iteration
    def _acc_name(self, name, j):
        # chunk 0 overwrites the flat gradient buffer, later chunks add to it: two different captured bodies
        return name if j == 0 else name + "/acc"

    def _fold(self):
        # fold: the small reductions ride INSIDE the optimizer graphs (G's 17.5 MB, D's 20 MB "rest", R1's) -- every graph
        # launch and every cross-stream join costs the GPU a dependency bubble (one rank on RCCL: 8 graphs + 3 joins per
        # iteration ran 3.6 % behind the 4-graph plain step); only D's head reduction (134 MB under the trunk's backward)
        # keeps its side stream.  Eager / gloo runs keep the asynchronous form.
        return (parallel.is_dist() and self.use_graphs and self.device.type == "cuda"
                and torch.distributed.get_backend() == "nccl" and os.environ.get("DGV2_NO_FOLDED_REDUCE") is None)

    def _g_substep(self, iteration, nacc, late_reals, log, real):
        """The G step in its exchanging / accumulating form: bodies, gradient reduction, optimizer step."""
        for j in gradient_accumulation(nacc, parallel.is_dist(), self.ddp_models):
            if j == 0 and not self._g_bufs_synced:
                parallel.sync_buffers(self.G)
            if self.use_real_in_g:
                log(self._run(self._acc_name("g_fb", j), self.g_fb_rel, real(j), j))
            else:
                log(self._run(self._acc_name("g_fb", j), self.g_fb, j))
        # G's 17.5 MB (+ rank 0's buffers) leave on the communication stream; the real batch of this iteration is
        # fetched / generated and converted meanwhile (the D step itself starts with a forward of the UPDATED G)
        fold = self._fold()

        def g_reduce_opt(sc):
            self.g_sync.all_reduce(carry=True)
            self._opt_step(self.optim_G)

        if fold:
            if late_reals:
                self.fetch_reals(next(self.iter_train_loader), out=self.x_real)
            # (its own body name: "g_opt" is the Adam-only graph of the non-folded form and of the path-length step)
            ok = self.g_sync.carry_ok()   # outside the graph: a replay carries what it carried at capture time
            self._run("g_red_opt", g_reduce_opt)
            self._g_bufs_synced = ok and self.g_sync.last_carried
        else:
            ok = self.g_sync.carry_ok()
            h = self.g_sync.all_reduce_captured(carry=True)
            if late_reals:
                self.fetch_reals(next(self.iter_train_loader), out=self.x_real)
            self.g_sync.wait(h)
            self._g_bufs_synced = h is not None and ok and self.g_sync.last_carried
            self._run("g_opt", lambda sc: self._opt_step(self.optim_G))


    def step(self, iteration):
        self.G.train()
        self.set_warmup_params(iteration)
        nacc = self.num_accumulation
        per_chunk = defaultdict(list)

        def log(sc):
            for k, v in sc.items():
                # under graph replay the body's scalar tensors are static buffers: keep a copy per chunk
                per_chunk[k].append(v.clone() if nacc > 1 else v)

        # a rank's share of the global batch, chunk by chunk (reference: trainer.py:253-257; the loops below run under
        # gans.context_manager.gradient_accumulation like the reference's: every chunk but the last inside the
        # gradient exchangers' no_sync, so that an exchange issued from inside a body would be suppressed there).  With one chunk and an
        # objective whose G step does not look at reals, the batch is prepared AFTER the G step's backward: it is the one
        # piece of work on this rank that does not depend on G's reduced gradient, so G's all-reduce runs under it.
        late_reals = nacc == 1 and not self.use_real_in_g
        reals = None
        if nacc > 1:
            reals = [self.fetch_reals(next(self.iter_train_loader))["image"] for _ in range(nacc)]
        elif not late_reals:
            self.fetch_reals(next(self.iter_train_loader), out=self.x_real)

        def real(j):
            if nacc > 1:
                self.x_real.copy_(reals[j])
            return self.x_real

        # DDP(broadcast_buffers=True) re-broadcasts rank 0's buffers before a forward of the wrapped G whenever the
        # PREVIOUS forward ran with grad mode on and outside `no_sync` (torch DDP: require_forward_param_sync).  The
        # reference's D step calls G with grad mode on (trainer.py:379, only requires_grad is off), so both G forwards
        # of an iteration are preceded by a broadcast -- between the two this rank's forward has moved ema_var / w_avg,
        # the second one is not redundant -- while the later chunks of an accumulation loop (their predecessors ran
        # under no_sync) are not.  Exactly that is kept.
        # Where the broadcast travels: the one before the G step rode in the previous iteration's tail exchange
        # (_g_bufs_synced), the one before the D step rides behind G's gradients; the stand-alone sync_buffers remains
        # for the first iteration and around the path-length step.
        # one process, one chunk: forward / backward and optimizer step of a sub-step are ONE body (see g_fb_opt)
        fuse_opt = (not parallel.is_dist()) and nacc == 1 and os.environ.get("DGV2_NO_FUSED_OPT") is None
        if fuse_opt:
            if self.use_real_in_g:
                log(self._run("g_step", self.g_fb_rel_opt, real(0), 0))
            else:
                log(self._run("g_step", self.g_fb_opt, 0))
                self.fetch_reals(next(self.iter_train_loader), out=self.x_real)
            fold = False
        else:
            self._g_substep(iteration, nacc, late_reals, log, real)
            fold = self._fold()

        if self.pl_weight > 0.0 and iteration % self.lazy_pl == 0:
            for j in gradient_accumulation(nacc, parallel.is_dist(), self.ddp_models):
                if j == 0 and not self._g_bufs_synced:
                    parallel.sync_buffers(self.G)
                log(self._run(self._acc_name("pl_fb", j), self.pl_fb, j))
            ok = self.g_sync.carry_ok()
            h = self.g_sync.all_reduce_captured(carry=True)
            self.g_sync.wait(h)
            self._g_bufs_synced = h is not None and ok and self.g_sync.last_carried
            self._run("g_opt", lambda sc: self._opt_step(self.optim_G))   # Adam only: the reduction ran just above

        fuse_d = fuse_opt and not self.split_d
        pending = []
        for j in gradient_accumulation(nacc, parallel.is_dist(), self.ddp_models):
            if fuse_d:
                log(self._run("d_step", self.d_fb_opt, real(j), j))
                continue
            if j == 0 and not self._g_bufs_synced:
                parallel.sync_buffers(self.G)
            if self.split_d:
                head, tail = self._acc_name("d_fb_head", j), self._acc_name("d_fb_tail", j)
                log(self._run(head, self.d_fb_head, real(j), j))
                if j == nacc - 1:
                    # 134 of D's 154 MB leave now, on the communication stream, under the trunk's backward
                    pending.append(self.d_sync.all_reduce_captured(part="first"))
                self._run(tail, self.d_fb_tail, j)
                self._link_graphs(head, tail)
            else:
                log(self._run(self._acc_name("d_fb", j), self.d_fb, real(j), j))
        # the (rest of the) gradient reduction of D runs on the communication stream while the EMA generator is updated
        # (G is final for this iteration: nothing below touches it)
        rest = "rest" if self.split_d else None
        if not fold and not fuse_d:
            pending.append(self.d_sync.all_reduce_captured(part=rest))
        decay = self.ema_decay(iteration)
        ema_inplace(self.G_ema, self.G, decay)
        for h in pending:
            self.d_sync.wait(h)

        def d_reduce_opt(part):
            def body(sc):
                self.d_sync.all_reduce(part=part)
                self._opt_step(self.optim_D)
            return body

        if fuse_d:
            pass                               # the optimizer step rode in the body
        elif fold:
            self._run("d_opt", d_reduce_opt(rest))
        else:
            self._run("d_opt", lambda sc: self._opt_step(self.optim_D))
        self._d_bank_fresh = False

        self._g_bufs_synced = False   # the D step's G forward moved this rank's ema_var / w_avg again
        r1_pending = None
        if self.gp_weight > 0.0 and iteration % self.lazy_gp == 0:
            for j in gradient_accumulation(nacc, parallel.is_dist(), self.ddp_models):
                if fuse_d:
                    log(self._run("r1_step", self.r1_fb_opt, real(j), j))
                else:
                    log(self._run(self._acc_name("r1_fb", j), self.r1_fb, real(j), j))
            # R1's 154 MB leave asynchronously; the scalar bookkeeping below runs under them.  The optimizer step cannot
            # move past the next iteration's G step, whose D forward must see the regularised weights (trainer.py:419-451).
            if not fold and not fuse_d:
                r1_pending = self.d_sync.all_reduce_captured()

        scalars = {k: (v[0] if len(v) == 1 else torch.stack([t.reshape(()) for t in v]).mean())
                   for k, v in per_chunk.items()}          # mean over the chunks (reference: trainer.py:471-476)
        # ONE small collective closes the iteration: the logged scalars, ADA's statistic pair when its update is due
        # (adaptive_augment.py:372-384) and rank 0's G buffers for the next iteration's first forward
        ada_due = iteration % self.lazy_ada == 0

        def tail(sc):
            res, ada_stats = parallel.tail_exchange(scalars, self.A.stats() if ada_due else None, self.G)
            if ada_due:
                res["stats/ada_rt"] = self.A.update_p(stats=ada_stats).reshape(())
                res["stats/ada_p"] = self.A.p.detach().clone()
            sc.update(res)

        # One communicator, one order: the tail exchange below is a second collective of the SAME RCCL communicator,
        # issued from the main stream (as a graph replay) or from c10d's stream (eager) -- neither is ordered against
        # a reduction replaying on the side stream, and two unordered collectives of one communicator can interleave
        # differently per rank (hang / corrupted buffers).  Join the reduction first (parallel._side_stream's invariant).
        if r1_pending is not None:
            self.d_sync.wait(r1_pending)
            r1_pending = None
        gp_key = "loss/D/gradient_penalty"
        captured_tail = (parallel.is_dist() and nacc == 1 and self.use_graphs and self.device.type == "cuda"
                         and torch.distributed.get_backend() == "nccl")
        if captured_tail:
            # The packing, the collective and the unpacking replay as a hipGraph too; its inputs are the bodies' static
            # scalar buffers.  Two variants only (ADA's update due or not): the R1 scalar, present every lazy.gp-th
            # iteration, travels EVERY iteration from a buffer of its own, so that no new key set -- i.e. no capture --
            # turns up deep into a run (a capture costs milliseconds; bench.py's timed region met one).
            if gp_key in scalars:
                self._gp_scalar.copy_(scalars[gp_key].reshape(()))
            had_gp = gp_key in scalars
            scalars = dict(scalars)
            scalars[gp_key] = self._gp_scalar
            out = dict(self._run("tail" + ("/ada" if ada_due else ""), tail))
            if not had_gp:
                out.pop(gp_key, None)
        else:
            out = {}
            tail(out)
        self._g_bufs_synced = parallel.is_dist()
        if self.gp_weight > 0.0 and iteration % self.lazy_gp == 0 and not fuse_d:
            if fold:   # (with the split D the plain step's optimizer graph reduces the "rest" segment only: another body)
                self._run("d_opt" if rest is None else "d_all_opt", d_reduce_opt(None))
            else:
                self._run("d_opt", lambda sc: self._opt_step(self.optim_D))
        set_requires_grad(self.D, False)

        if self.sync_scalars:
            out = {k: v.cpu().item() for k, v in out.items()}
            self.check_status()
        out["stats/ema_decay"] = decay
        out["stats/warmup_blur_sigma"] = self.blur_sigma
        out["stats/warmup_dropout_ratio"] = self.dropout_ratio
        return out

    def check_status(self):
        """Read the kernels' status word (synchronises) and raise if a launch met a broken promise -- e.g. a value that is
        not bf16-representable in the channels Discriminator.forward promised to be (x_exact): the fp32 epilogue conv
        would then have computed on bf16-rounded features without any other sign.  Called where the host waits for the
        device anyway: the scalar read-back of step(), validation(), save_checkpoint()."""
        dgv2_native.status_check()

    def graphs_live(self):
        """name -> True (captured, replaying) / False (capture failed: that body runs eagerly).  bench.py asserts that
        no body silently fell back."""
        return {k: v is not None for k, v in self._graphs.items()}

    @torch.no_grad()
    def sample(self, ema=False):
        model = self.G_ema if ema else self.G
        model.eval()
        return model(self.z_fixed, **self.auxin)

    @torch.no_grad()
    def pointnet_features(self, depth, pointnet):
        """Generator / dataset output in [-1, 1] -> PointNet features of the normalised point cloud
        (reference: get_pointnet_features, trainer.py:505-510)."""
        depth = tanh_to_sigmoid(depth).clamp(0, 1)
        points = self.coord.convert(depth, "inv_depth_norm", "point_set")
        points = points / self.coord.max_depth
        return pointnet(points.transpose(1, 2))

    @torch.no_grad()
    def validation(self, num_fakes=10_000, pointnet=None, max_reals=None):
        """Feature-based validation scores (reference: trainer.py:495-549): Frechet and squared-MMD distances between
        PointNet features of `num_fakes` EMA-generator samples and of the training set (features cached on the host
        after the first call).  The features are computed on this rank's GPU, the two distances on the host in float64
        like the reference.  `pointnet`: the feature extractor (default: pretrained_pointnet(), which needs
        cls_model_39.pth on disk); `max_reals` bounds the real-data pass (the synthetic dataset is endless: it
        contributes num_fakes samples)."""
        from gans.metrics.fpd_kpd import compute_frechet_distance, compute_squared_mmd
        from gans.metrics.pointnet import pretrained_pointnet
        N = int(num_fakes)
        B = self.B
        net = (pretrained_pointnet() if pointnet is None else pointnet).to(self.device)
        self.G_ema.eval()

        if getattr(self, "val_real_feats", None) is None:
            feats = []
            if self.train_dataset is not None:   # one pass over the files, in order, rank-local like the reference
                n_real = len(self.train_dataset) if max_reals is None else min(len(self.train_dataset), int(max_reals))
                for lo in range(0, n_real, B):
                    items = [self.train_dataset[i] for i in range(lo, min(lo + B, n_real))]
                    batch = {k: torch.stack([it[k] for it in items]) for k in ("depth", "mask")}
                    feats.append(self.pointnet_features(self.fetch_reals(batch)["image"], net).cpu())
            else:
                n_real = N if max_reals is None else int(max_reals)
                for lo in range(0, n_real, B):
                    x = self.fetch_reals(next(self.iter_train_loader))["image"][:n_real - lo]
                    feats.append(self.pointnet_features(x, net).cpu())
            self.val_real_feats = torch.cat(feats, dim=0)

        fake = []
        for lo in range(0, N, B):
            bs = min(B, N - lo)
            out = self.G_ema(self.sample_z(bs), **self.auxin)
            fake.append(self.pointnet_features(out["image"], net).cpu())
        fake = torch.cat(fake, dim=0)

        self.check_status()
        tag = f"{N // 1000}k"
        f1, f2 = fake.double().numpy(), self.val_real_feats.double().numpy()
        return {
            f"pointcloud/frechet_distance_{tag}": compute_frechet_distance(feats1=f1, feats2=f2),
            f"pointcloud/squared_mmd_{tag}": compute_squared_mmd(feats1=f1, feats2=f2),
        }

    def save_checkpoint(self, save_path, step):
        """Same keys as the reference (trainer.py:551-567); `cfg` is stored as the reference stores it (an OmegaConf node)
        where omegaconf is importable, as plain containers otherwise (see `plain`)."""
        def optim_state(opt):
            # the fused Adam kernel keeps ONE device step counter that every per-parameter `step` entry views; a
            # checkpoint carries independent copies so that stock torch.optim.Adam can resume from it
            sd = opt.state_dict()
            for st in sd["state"].values():
                if torch.is_tensor(st.get("step")):
                    st["step"] = st["step"].detach().clone().reshape(())
            return sd

        def plain(o):
            # cfg as plain dicts / lists: loadable anywhere (weights_only=True) and by gans.pretrained.load_checkpoint,
            # which turns it back into an attribute-accessible Config.  NOT directly by the reference's own tools: they
            # index ckpt["cfg"].model.generator... by attribute (quick_demo.py:25, test_gan.py:48) because the reference
            # pickles its OmegaConf node (trainer.py:551-567); omegaconf is not a dependency here, so a file meant for
            # upstream code goes through gans.pretrained.to_upstream / scripts/ckpt_to_upstream.py (OmegaConf.create)
            # in an environment that has it
            if isinstance(o, dict):
                return {k: plain(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [plain(v) for v in o]
            return o

        cfg_out = plain(self.cfg)
        try:   # the reference pickles its OmegaConf node (trainer.py:551-567) and its tools read it by attribute
            # (quick_demo.py:25, test_gan.py:48): where omegaconf is importable the file is written in that form and is
            # directly consumable upstream; gans.pretrained.load_checkpoint accepts both forms
            from omegaconf import OmegaConf
            cfg_out = OmegaConf.create(cfg_out)
        except ImportError:
            pass
        ckpt = {
            "cfg": cfg_out, "step": step, "angle": self.coord.angle.detach().cpu(),
            "G": self.G.state_dict(), "D": self.D.state_dict(), "G_ema": self.G_ema.state_dict(),
            "A": self.A.state_dict(), "optim_G": optim_state(self.optim_G), "optim_D": optim_state(self.optim_D),
        }
        if self.pl_weight > 0.0:
            ckpt["pl_ema"] = self.pl_ema.detach().cpu()
        if self.native_rng:
            # one key more than the reference writes (it restores no RNG state): (seed, offset, ticket, 0) of
            # the device-side Philox stream, so that a resumed run does not replay the draws of its first iterations
            from gans.models.ops import native as _native
            ckpt["rng_state"] = _native.rng_state(self.device).detach().cpu()
        self.check_status()   # (state_dict -> file synchronises anyway) never write weights trained on a broken promise
        save_path.parent.mkdir(parents=True, exist_ok=True)
        torch.save(ckpt, save_path)
