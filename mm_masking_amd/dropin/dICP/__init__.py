"""Drop-in for the absent third-party package `dICP` (`from dICP.ICP import ICP`)."""
