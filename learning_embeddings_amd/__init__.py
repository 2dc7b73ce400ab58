"""learning_embeddings_amd -- MI355X-native joint image+label hyperbolic entailment-cone training path.

Host side: Python classes with the reference's names and signatures (network/oe_h.py, order_embeddings.py, loss.py,
experiment.py, embed_toy.py of ankitdhall/learning_embeddings).  Device side: liblecone.so (hand-written HIP for gfx950
+ a bit-exact host sampler) behind the C ABI in include/lecone.h.  Importing this package loads the library and fails
loudly if it has not been built -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when liblecone.so is missing or stale)

__all__ = ['_lib']
