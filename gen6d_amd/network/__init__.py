"""Registry with the reference's names (network/__init__.py:5-9): `name2network[cfg['network']](cfg)`."""
from .detector import Detector
from .refiner import VolumeRefiner
from .selector import ViewpointSelector

name2network = {
    "refiner": VolumeRefiner,
    "detector": Detector,
    "selector": ViewpointSelector,
}
