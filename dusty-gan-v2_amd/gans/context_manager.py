"""Gradient-accumulation helper (reference: gans/context_manager.py:21-35): every micro-step but the
last runs with gradient exchange suppressed.  Works with torch DDP (`no_sync`) and with this
repo's FlatGradSync (gans/parallel.py) which exposes the same `no_sync` context; Trainer.step drives its
chunk loops through it."""
from contextlib import ExitStack, contextmanager


@contextmanager
def _noop():
    yield


def gradient_accumulation(num_accumulation, is_ddp, ddp_models):
    for i in range(num_accumulation):
        last = i == num_accumulation - 1
        with ExitStack() as stack:
            if is_ddp and not last:
                for m in ddp_models:
                    stack.enter_context(m.no_sync())
            else:
                stack.enter_context(_noop())
            yield i
