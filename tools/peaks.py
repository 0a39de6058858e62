import sys; sys.path.insert(0,'/root/repo')
from ddmi_amd import _lib
_lib.init_device(0)
print(_lib.measured_peaks())
