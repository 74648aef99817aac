"""dspnet_amd: MI355X (gfx950) native hot path of DSPNet -- the conv-heavy multi-task
forward/backward and the SSD multibox operators -- behind the reference's operator API."""
__version__ = "0.1.0"
