"""Stand-in for the un-installed third-party `torchvision` -- container-only test
infrastructure used by oracle/ref_import.py.  Exposes just what the reference
imports: models.densenet121 and transforms.functional.pad."""
from . import models, transforms  # noqa: F401
