"""Variable-name constants the driver imports (reference: data/processing/era5_constants.py)."""
PRESSURE_LEVEL_VARS = ["geopotential", "u_component_of_wind", "v_component_of_wind", "temperature",
                       "relative_humidity", "specific_humidity", "vorticity", "potential_vorticity"]
DEFAULT_PRESSURE_LEVELS = [50, 250, 500, 600, 700, 850, 925]
CONSTANTS = ["orography", "land_sea_mask", "slt", "lattitude", "longitude"]
PRECIP_VARIABLES = ["total_precipitation_24hr"]
