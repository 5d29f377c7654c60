"""easynlp_amd -- MI355X (gfx950) native CLIP text-image retrieval hot path behind
the EasyNLP AppZoo API.  Product code: HIP kernels + C ABI in
``csrc/`` (``libezclip_hip.so``), ctypes binding in ``lib.py``, the host-side
mirror of the reference application in ``appzoo/clip``."""
__version__ = "0.1.0"
