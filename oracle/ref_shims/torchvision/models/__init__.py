from oracle.densenet import densenet121, DenseNet121  # noqa: F401
