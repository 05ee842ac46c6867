"""zsgnet-pytorch_amd — MI355X-native ZSGNet training-step hot path.

The directory name carries a hyphen (the task's required layout); import it as `zsgnet_pytorch_amd` (an alias package
at the repo root points here).  Importing any compute module loads libzsg.so and fails loudly if it is missing."""
__version__ = "0.1.0"
