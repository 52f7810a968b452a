"""Drop-in modules for the reference's `model` package: PoseNet, TrajNet (+ControlNet), their heads and the
evaluation loss reports.  The `nn.Module`s hold parameters under the reference's keys; forwards run in HIP."""
