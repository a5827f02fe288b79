"""``python -m fastspeech2_lightning_amd train CONFIG.yaml ...`` (the ``fs2l`` entry)."""
from .cli import main

raise SystemExit(main())
