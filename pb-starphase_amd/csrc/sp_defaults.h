// sp_defaults.h -- the configuration a database file without "cyp2d6_config" / "hla_config" gets: the values of Cyp2d6Config::default()
// (src/cyp2d6/definitions.rs:128-301) and HlaConfig's default HLA-A / HLA-B coordinates (src/hla/alleles.rs:232-318), written as the
// JSON objects a database file carries (0-based half-open coordinates; data only).
#pragma once
static const char SP_DEFAULT_CYP2D6_CONFIG[] =
    "{\"cyp2d6_star5_del\":{\"chrom\":\"chr22\",\"end\":42135343,\"start\":42123191},\"cyp_coordinates\":{\"CYP2D6\":{\"chrom\":\"chr22\",\"end\":42132424,\"start\":42126259},\"C"
    "YP2D6_wfa_backbone\":{\"chrom\":\"chr22\",\"end\":42132424,\"start\":42126259},\"CYP2D7\":{\"chrom\":\"chr22\",\"end\":42145903,\"start\":42139965},\"REP6\":{\"chrom\":\"chr2"
    "2\",\"end\":42125963,\"start\":42123191},\"REP7\":{\"chrom\":\"chr22\",\"end\":42138115,\"start\":42135343},\"link_region\":{\"chrom\":\"chr22\",\"end\":42135343,\"start\":421"
    "32424},\"spacer\":{\"chrom\":\"chr22\",\"end\":42139679,\"start\":42138115}},\"cyp_regions\":{\"CYP2D6\":{\"exon1\":{\"chrom\":\"chr22\",\"end\":42130810,\"start\":42130611},"
    "\"exon2\":{\"chrom\":\"chr22\",\"end\":42129909,\"start\":42129737},\"exon3\":{\"chrom\":\"chr22\",\"end\":42129185,\"start\":42129032},\"exon4\":{\"chrom\":\"chr22\",\"end\":421"
    "28944,\"start\":42128783},\"exon5\":{\"chrom\":\"chr22\",\"end\":42128350,\"start\":42128173},\"exon6\":{\"chrom\":\"chr22\",\"end\":42127983,\"start\":42127841},\"exon7\":{\""
    "chrom\":\"chr22\",\"end\":42127634,\"start\":42127446},\"exon8\":{\"chrom\":\"chr22\",\"end\":42126992,\"start\":42126850},\"exon9\":{\"chrom\":\"chr22\",\"end\":42126752,\"sta"
    "rt\":42126498}},\"CYP2D7\":{\"exon1\":{\"chrom\":\"chr22\",\"end\":42144483,\"start\":42144283},\"exon2\":{\"chrom\":\"chr22\",\"end\":42143581,\"start\":42143409},\"exon3\":{"
    "\"chrom\":\"chr22\",\"end\":42142880,\"start\":42142727},\"exon4\":{\"chrom\":\"chr22\",\"end\":42142639,\"start\":42142478},\"exon5\":{\"chrom\":\"chr22\",\"end\":42142044,\"st"
    "art\":42141867},\"exon6\":{\"chrom\":\"chr22\",\"end\":42141675,\"start\":42141533},\"exon7\":{\"chrom\":\"chr22\",\"end\":42141339,\"start\":42141151},\"exon8\":{\"chrom\":\"c"
    "hr22\",\"end\":42140696,\"start\":42140554},\"exon9\":{\"chrom\":\"chr22\",\"end\":42140456,\"start\":42140202}}},\"cyp_translate\":{\"CYP2D6::CYP2D7::exon2\":\"68\",\"CYP2"
    "D6::CYP2D7::exon8\":\"61\",\"CYP2D6::CYP2D7::intron1\":\"68\",\"CYP2D6::CYP2D7::intron8\":\"63\",\"CYP2D7::CYP2D6::exon2\":\"13\",\"CYP2D7::CYP2D6::exon3\":\"13\",\"CYP2D"
    "7::CYP2D6::exon4\":\"13\",\"CYP2D7::CYP2D6::exon5\":\"13\",\"CYP2D7::CYP2D6::exon6\":\"13\",\"CYP2D7::CYP2D6::exon7\":\"13\",\"CYP2D7::CYP2D6::exon8\":\"13\",\"CYP2D7::CY"
    "P2D6::exon9\":\"13\",\"CYP2D7::CYP2D6::intron1\":\"13\",\"CYP2D7::CYP2D6::intron2\":\"13\",\"CYP2D7::CYP2D6::intron3\":\"13\",\"CYP2D7::CYP2D6::intron4\":\"13\",\"CYP2D7:"
    ":CYP2D6::intron5\":\"13\",\"CYP2D7::CYP2D6::intron6\":\"13\",\"CYP2D7::CYP2D6::intron7\":\"13\",\"CYP2D7::CYP2D6::intron8\":\"13\"},\"inferred_connections\":[[\"*1\",\"*1"
    "\"],[\"*10\",\"*10\"],[\"*10\",\"*36\"],[\"*146\",\"*146\"],[\"*17\",\"*17\"],[\"*2\",\"*2\"],[\"*28\",\"*28\"],[\"*29\",\"*29\"],[\"*3\",\"*3\"],[\"*35\",\"*35\"],[\"*4\",\"*4\"],[\"*4\",\"*68\""
    "],[\"*41\",\"*41\"],[\"*43\",\"*43\"],[\"*45\",\"*45\"],[\"*6\",\"*6\"],[\"*9\",\"*9\"]],\"unexpected_singletons\":[\"*36\",\"*68\"]}";
static const char SP_DEFAULT_HLA_CONFIG[] =
    "{\"hla_coordinates\":{\"HLA-A\":{\"chrom\":\"chr6\",\"end\":29945870,\"start\":29942253},\"HLA-B\":{\"chrom\":\"chr6\",\"end\":31357442,\"start\":31353361}},\"hla_exons\":{\"H"
    "LA-A\":[{\"chrom\":\"chr6\",\"end\":29942626,\"start\":29942531},{\"chrom\":\"chr6\",\"end\":29943026,\"start\":29942756},{\"chrom\":\"chr6\",\"end\":29943543,\"start\":299432"
    "67},{\"chrom\":\"chr6\",\"end\":29944397,\"start\":29944121},{\"chrom\":\"chr6\",\"end\":29944616,\"start\":29944499},{\"chrom\":\"chr6\",\"end\":29945091,\"start\":29945058}"
    ",{\"chrom\":\"chr6\",\"end\":29945281,\"start\":29945233},{\"chrom\":\"chr6\",\"end\":29945870,\"start\":29945450}],\"HLA-B\":[{\"chrom\":\"chr6\",\"end\":31354296,\"start\":31"
    "353874},{\"chrom\":\"chr6\",\"end\":31354526,\"start\":31354478},{\"chrom\":\"chr6\",\"end\":31354665,\"start\":31354632},{\"chrom\":\"chr6\",\"end\":31355223,\"start\":31355"
    "106},{\"chrom\":\"chr6\",\"end\":31355592,\"start\":31355316},{\"chrom\":\"chr6\",\"end\":31356442,\"start\":31356166},{\"chrom\":\"chr6\",\"end\":31356957,\"start\":31356687"
    "},{\"chrom\":\"chr6\",\"end\":31357179,\"start\":31357085}]},\"hla_is_forward_strand\":{\"HLA-A\":true,\"HLA-B\":false}}";
