""" Same import path as reference ``bild.util`` (bild/util.py); implementation in `profiles`. """
from .profiles import Loopingprofile, state_probabilities  # noqa: F401
