"""The run configurations live with the package (radiative3d_amd/configs.py) so that bench.py
does not depend on the test tree; tests keep importing them from here."""
from radiative3d_amd.configs import *  # noqa: F401,F403
from radiative3d_amd.configs import CONFIGS  # noqa: F401
