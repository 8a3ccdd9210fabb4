"""Model registry (pc/model/__init__.py:11-31): load_model(name) -> class."""
from . import res16unet

MODELS = [getattr(res16unet, a) for a in dir(res16unet) if "Net" in a and isinstance(getattr(res16unet, a), type)]


def get_models():
  return MODELS


def load_model(name):
  """Returns the model class registered under `name` (None and a listing if unknown)."""
  by_name = {m.__name__: m for m in MODELS}
  if name not in by_name:
    print("Invalid model index. Options are:")
    for m in MODELS:
      print("\t* {}".format(m.__name__))
    return None
  return by_name[name]
