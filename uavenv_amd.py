"""Import alias: `import uavenv_amd` loads the package that lives in the directory
`-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd/`
(a name Python cannot import directly: leading '-' and a '.')."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd")
_spec = importlib.util.spec_from_file_location("uavenv_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["uavenv_amd"] = _mod
_spec.loader.exec_module(_mod)
