import os, sys
ROOT="/root/repo"
sys.path.insert(0, ROOT+"/deflate-rs_amd"); sys.path.insert(0, ROOT+"/tests")
import datagen, deflate_amd as da
ctx = da.Context(0)
for name, data in (("text100", datagen.text_like(100_000_000, 0x656E77696B38)), ("silesia", datagen.silesia_like(0x53494C45)), ("webtext256", datagen.webtext(256<<20)), ("mixed64", datagen.mixed(64_000_000, 5))):
    for lv in (da.Compression.Default, da.Compression.Fast, da.Compression.Best):
        c = da.Context(0)
        c.encode(data, lv)
        i = c.info(); print(os.environ.get("MI355_DEFLATE_LIB","")[-12:], name, lv.name, "fallback", i["spec_fallback"], "repaired", i["spec_repaired"], flush=True)
        c.close()
