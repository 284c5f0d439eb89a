"""Python mirrors of the reference's Julia host code ABOVE the solver seams (test infrastructure only).

The reference's state machines (GMRFWorkspace, WorkspacePool, the LinearSolve cache protocol, ConstraintInfo) stay
the reference's own Julia; Julia is absent from this image, so the parity tests drive libgmrfx.so through these
restatements of their call sequences -- so that a test here reads like the reference test it reproduces."""
from .workspace import GMRFWorkspace, WorkspacePool  # noqa: F401
