"""xmhw_amd -- MI355X (gfx950) implementation of xmhw's threshold() hot path and of its consumer.

    from xmhw_amd import threshold, detect     # same signatures as xmhw.xmhw.threshold / detect

The device side is hand-written HIP behind a C ABI (include/xmhw_amd.h); this
package holds only the host side of the path.  Importing it does not need a
GPU; calling threshold() does, and fails loudly without the HIP extension.
"""
from .exception import XmhwException
from .api import threshold, threshold_array, GridSeries, ClimDataset
from .calendar import add_doy, get_calendar
from .landmask import land_check
from .detect import detect, threshold_detect, EventDataset, InterDataset, climatology_series
from .device import release_device_cache
from .stats import block_average, BlockDataset
from .ingest import open_series, threshold_file

__all__ = ["threshold", "threshold_array", "GridSeries", "ClimDataset", "XmhwException",
           "add_doy", "get_calendar", "land_check", "detect", "threshold_detect", "EventDataset", "InterDataset",
           "climatology_series", "release_device_cache", "block_average", "BlockDataset", "open_series",
           "threshold_file"]
__version__ = "0.1.0"
