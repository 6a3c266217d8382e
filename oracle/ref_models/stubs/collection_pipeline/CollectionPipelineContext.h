// oracle/ref_models/stubs -- TEST INFRASTRUCTURE.  Stands in for core/collection_pipeline/CollectionPipelineContext.h (queues, alarm
// manager, logger, flusher map ...: none of it compiles here) so that the REFERENCE's plugin/processor/CommonParserOptions.cpp can be
// compiled from where it lies.  That file touches the context only inside its PARAM_WARNING_* macros (stubs/common/ParamExtractor.h).
#pragma once
#include <memory>
#include <string>

#include "models/PipelineEventGroup.h"  // GroupMetadata, EventGroupMetaKey (what the real header brings in)

namespace logtail {
class CollectionPipelineContext {
public:
    const std::string& GetConfigName() const { return mEmpty; }
    const std::string& GetProjectName() const { return mEmpty; }
    const std::string& GetLogstoreName() const { return mEmpty; }
    const std::string& GetRegion() const { return mEmpty; }

private:
    std::string mEmpty;
};
}  // namespace logtail
